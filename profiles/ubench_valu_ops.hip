// Issue cost of the VALU instructions the extz2 kernels are made of, per wave-instruction per SIMD, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu_ops ubench_valu_ops.hip && ./ubench_valu_ops
// Eight independent dependency chains per wavefront, 4 and 8 wavefronts a SIMD, every SIMD of the device busy; the figure is
// elapsed time x 2.4 GHz / (instructions of one wavefront x wavefronts per SIMD).  (profiles/r05_ubench_exec_mask.txt found
// v_add_u32 at 2.4 and a mix of v_pk_* at 4.2 cycles: which of the other encodings and data paths are on which side?)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
// one op over the eight chains: %0..%7 accumulators, %8 a second vector operand, %9 a third
#define OP8(fmt)                                                                                                          \
  REP16(asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                                               \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
                     : "v"(b), "v"(c), "s"(sm)                                                                             \
                     : "vcc");)
#define S(i) #i
#define F2(op, i) op " %" S(i) ", %" S(i) ", %8\n"
#define F3(op, i) op " %" S(i) ", %" S(i) ", %8, %9\n"

#define D_pk_add(i) F2("v_pk_add_u16", i)
#define D_pk_sub(i) F2("v_pk_sub_i16", i)
#define D_pk_maxi(i) F2("v_pk_max_i16", i)
#define D_pk_minu(i) F2("v_pk_min_u16", i)
#define D_pk_mad(i) F3("v_pk_mad_u16", i)
#define D_pk_shl(i) F2("v_pk_lshlrev_b16", i)
#define D_add32(i) F2("v_add_u32", i)
#define D_sub32(i) F2("v_sub_u32", i)
#define D_and(i) F2("v_and_b32", i)
#define D_or(i) F2("v_or_b32", i)
#define D_maxi32(i) F2("v_max_i32", i)
#define D_minu32(i) F2("v_min_u32", i)
#define D_shl32(i) F2("v_lshlrev_b32", i)
#define D_cnd_vcc(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %8, vcc\n"
#define D_cnd_sgpr(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %8, %10\n"
#define D_lshl_or(i) "v_lshl_or_b32 %" S(i) ", %" S(i) ", 1, %8\n"
#define D_lshl_add(i) "v_lshl_add_u32 %" S(i) ", %" S(i) ", 1, %8\n"
#define D_and_or(i) F3("v_and_or_b32", i)
#define D_bfi(i) F3("v_bfi_b32", i)
#define D_alignbit(i) "v_alignbit_b32 %" S(i) ", %" S(i) ", %8, 16\n"
#define D_perm(i) F3("v_perm_b32", i)
#define D_mov_dpp(i) "v_mov_b32_dpp %" S(i) ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define D_mov_dpp_row(i) "v_mov_b32_dpp %" S(i) ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define D_add_dpp(i) "v_add_u32_dpp %" S(i) ", %8, %" S(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define D_add16(i) F2("v_add_u16", i)
#define D_maxi16(i) F2("v_max_i16", i)
#define D_add16_sdwa(i) "v_add_u16_sdwa %" S(i) ", %" S(i) ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define D_max3i32(i) F3("v_max3_i32", i)
#define D_max3i16(i) F3("v_max3_i16", i)
#define D_add3(i) F3("v_add3_u32", i)
#define D_mad24(i) F3("v_mad_u32_u24", i)
#define D_mul24(i) F2("v_mul_u32_u24", i)
#define D_mov(i) "v_mov_b32 %" S(i) ", %8\n"
#define D_cmp_vcc(i) "v_cmp_gt_i32 vcc, %" S(i) ", %8\n"
#define D_cmp_addc(i) "v_cmp_gt_i32 vcc, %" S(i) ", %8\n v_addc_co_u32 %" S(i) ", vcc, %" S(i) ", %" S(i) ", vcc\n"
#define D_addc(i) "v_addc_co_u32 %" S(i) ", vcc, %" S(i) ", %8, vcc\n"
#define D_add_sgpr(i) "v_add_u32 %" S(i) ", s20, %" S(i) "\n"
#define D_add_lit(i) "v_add_u32 %" S(i) ", 0x12345, %" S(i) "\n"
#define D_pk_add_lit(i) "v_pk_add_u16 %" S(i) ", %" S(i) ", 1 op_sel_hi:[1,0]\n"
#define D_xor(i) F2("v_xor_b32", i)
#define D_sad_u8(i) F3("v_sad_u8", i)
#define D_fma32(i) "v_fma_f32 %" S(i) ", %" S(i) ", %8, %9\n"
#define D_fmac32(i) "v_fmac_f32 %" S(i) ", %8, %9\n"
#define D_sub_e64(i) "v_sub_u32_e64 %" S(i) ", %" S(i) ", %8\n"
#define D_add_e64_clamp(i) "v_add_u32_e64 %" S(i) ", %" S(i) ", %8 clamp\n"
#define D_max_u16(i) F2("v_max_u16", i)
#define D_lshrrev(i) F2("v_lshrrev_b32", i)
#define D_ashrrev(i) F2("v_ashrrev_i32", i)
#define D_bfe(i) "v_bfe_u32 %" S(i) ", %" S(i) ", 8, 8\n"
#define D_readlane(i) "v_readlane_b32 s22, %" S(i) ", 5\n"
#define D_writelane(i) "v_writelane_b32 %" S(i) ", s20, 5\n"

#define OPS(X)                                                                                                                          \
  X(pk_add) X(pk_sub) X(pk_maxi) X(pk_minu) X(pk_mad) X(pk_shl) X(pk_add_lit) X(add32) X(sub32) X(sub_e64) X(add_e64_clamp) X(and) X(or) \
  X(xor) X(maxi32) X(minu32) X(shl32) X(lshrrev) X(ashrrev) X(bfe) X(cnd_vcc) X(cnd_sgpr) X(lshl_or) X(lshl_add) X(and_or) X(bfi)       \
  X(alignbit) X(perm) X(mov) X(mov_dpp) X(mov_dpp_row) X(add_dpp) X(add16) X(maxi16) X(max_u16) X(add16_sdwa) X(max3i32) X(max3i16)     \
  X(add3) X(mad24) X(mul24) X(cmp_vcc) X(addc) X(cmp_addc) X(add_sgpr) X(add_lit) X(sad_u8) X(fma32) X(fmac32) X(readlane) X(writelane)

enum {
#define X(n) OP_##n,
  OPS(X)
#undef X
      OP_COUNT
};
static const char *kNames[] = {
#define X(n) #n,
    OPS(X)
#undef X
};
static const int kPer[] = {  // instructions per D_ macro
#define X(n) (OP_##n == OP_cmp_addc ? 2 : 1),
    OPS(X)
#undef X
};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned long long sm) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9, a6 = a0 * 11, a7 = a0 + 77,
           b = blockIdx.x + 12345, c = threadIdx.x * 0x01010101u;
  asm volatile("s_mov_b32 s20, 77" ::: "s20", "s22");
  for (int i = 0; i < iters; i++) {
#define X(n) \
  if (OP == OP_##n) { OP8(D_##n) }
    OPS(X)
#undef X
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(unsigned *d) {
  double cyc[2];
  int wi = 0;
  for (int waves_per_simd : {4, 8}) {
    const int iters = 1000, blocks = 256 * 4 * waves_per_simd / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 0x5555555555555555ull);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      k<OP><<<blocks, 256>>>(d, iters, 0x5555555555555555ull);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double instr = (double)iters * 16 * 8 * kPer[OP];
    cyc[wi++] = best * 1e6 / (instr * waves_per_simd) * 2.4;
  }
  printf("%-16s %6.2f (4 waves/SIMD) %6.2f (8 waves/SIMD) cycles per wave-instruction per SIMD at 2.4 GHz\n", kNames[OP], cyc[0], cyc[1]);
}
template <int OP>
void run_all(unsigned *d) {
  if constexpr (OP < OP_COUNT) {
    run<OP>(d);
    run_all<OP + 1>(d);
  }
}
int main() {
  unsigned *d;
  hipMalloc(&d, 256 * 4 * 8 * 64 * 4 * 4);
  run_all<0>(d);
  return 0;
}
