#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template<int OP> __global__ void k(unsigned* out, int iters) {
  unsigned a0=threadIdx.x, a1=a0*3+1, a2=a0*5+2, a3=a0*7+3, a4=a0^0x55, a5=a0+9, a6=a0*11, a7=a0+77, b=blockIdx.x+12345;
  for (int i=0;i<iters;i++) {
    if (OP==0) { REP16( asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==1) { REP16( asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==2) { REP16( asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==3) { REP16( asm volatile("v_pk_mad_u16 %0, %0, %8, %8\n v_pk_mad_u16 %1, %1, %8, %8\n v_pk_mad_u16 %2, %2, %8, %8\n v_pk_mad_u16 %3, %3, %8, %8\n v_pk_mad_u16 %4, %4, %8, %8\n v_pk_mad_u16 %5, %5, %8, %8\n v_pk_mad_u16 %6, %6, %8, %8\n v_pk_mad_u16 %7, %7, %8, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==4) { REP16( asm volatile("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==5) { REP16( asm volatile("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==6) { REP16( asm volatile("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==7) { REP16( asm volatile("v_cndmask_b32_sdwa %0, %0, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %1, %1, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %2, %2, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %3, %3, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %4, %4, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %5, %5, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %6, %6, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32_sdwa %7, %7, %8, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b) : "vcc"); ) }
    if (OP==8) { REP16( asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==9) { REP16( asm volatile("v_pk_sub_i16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_sub_i16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %8\n v_pk_sub_i16 %5, %5, %8\n v_pk_max_u16 %6, %6, %8\n v_pk_sub_i16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b)); ) }
    if (OP==10) { REP16( asm volatile("s_add_i32 s20, s20, 1\n s_add_i32 s21, s21, 1\n s_add_i32 s22, s22, 1\n s_add_i32 s23, s23, 1\n s_add_i32 s24, s24, 1\n s_add_i32 s25, s25, 1\n s_add_i32 s26, s26, 1\n s_add_i32 s27, s27, 1" ::: "s20","s21","s22","s23","s24","s25","s26","s27","scc"); ) }
    if (OP==11) { REP16( asm volatile("v_add_u32 %0, %0, %8\n s_add_i32 s20, s20, 1\n v_add_u32 %1, %1, %8\n s_add_i32 s21, s21, 1\n v_add_u32 %2, %2, %8\n s_add_i32 s22, s22, 1\n v_add_u32 %3, %3, %8\n s_add_i32 s23, s23, 1" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3) : "v"(a4),"v"(a5),"v"(a6),"v"(a7),"v"(b) : "s20","s21","s22","s23","scc"); ) }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7;
}
template<int OP> double run(int waves_per_simd, const char* name) {
  unsigned* d; hipMalloc(&d, 256*4*8*64*4*4);
  int iters=2000; int blocks=256*4*waves_per_simd/4;   // 256-thread blocks: 4 waves each
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks,256>>>(d,10);
  hipEventRecord(e0); k<OP><<<blocks,256>>>(d,iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double instr_per_wave=(double)iters*16*8;
  double waves_per_simd_total = (double)blocks*4/(256*4);
  double ns_per_instr_per_simd = ms*1e6/(instr_per_wave*waves_per_simd_total);
  printf("%-28s waves/SIMD=%d  %.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ns_per_instr_per_simd, ns_per_instr_per_simd*2.4);
  hipFree(d); return ns_per_instr_per_simd;
}
int main(){
  for (int w : {1,4,8}) {
    run<0>(w,"v_add_u32"); run<1>(w,"v_pk_add_u16"); run<2>(w,"v_pk_max_i16"); run<3>(w,"v_pk_mad_u16");
    run<4>(w,"v_mov_dpp wave_shr"); run<5>(w,"v_mov_dpp row_shr"); run<6>(w,"v_perm_b32"); run<7>(w,"v_cndmask_sdwa");
    run<8>(w,"v_lshl_add_u32"); run<9>(w,"pk mix"); run<10>(w,"s_add_i32"); run<11>(w,"valu+salu interleaved(8)");
  }
}
