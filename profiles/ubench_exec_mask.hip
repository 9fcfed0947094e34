// Does a packed 16-bit VALU instruction with part of EXEC off issue faster on gfx950?  (The headline pair kernel's window of
// 144 slots is 2.25 registers: its third register row has 16 useful lanes.)   hipcc --offload-arch=gfx950 -O3 ubench_exec_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(unsigned *out, int iters, unsigned long long mask) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9, a6 = a0 * 11, a7 = a0 + 77,
           b = blockIdx.x + 12345;
  const unsigned lane = threadIdx.x & 63;
  if ((mask >> lane) & 1ull) {  // EXEC = mask for the whole loop
    for (int i = 0; i < iters; i++) {
      if (OP == 0) { REP16(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_sub_i16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_max_u16 %5, %5, %8\n v_pk_sub_i16 %6, %6, %8\n v_pk_mad_u16 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
      if (OP == 1) { REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
void run(int waves_per_simd, unsigned long long mask, const char *name) {
  unsigned *d;
  hipMalloc(&d, 256 * 4 * 8 * 64 * 4 * 4);
  int iters = 2000, blocks = 256 * 4 * waves_per_simd / 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 10, mask);
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, iters, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)iters * 16 * 8, waves = (double)blocks * 4 / (256 * 4);
  printf("%-14s EXEC %016llx  waves/SIMD %d  %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, mask, waves_per_simd,
         ms * 1e6 / (instr * waves) * 2.4);
  hipFree(d);
}
int main() {
  for (int w : {1, 4, 8})
    for (unsigned long long m : {~0ull, 0xffffffffull, 0xffffull, 0xffff0000ffffull, 0xfull, 0xffffffff00000000ull}) {
      run<0>(w, m, "v_pk_* mix");
      run<1>(w, m, "v_add_u32");
    }
}
