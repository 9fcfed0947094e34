// Internal types shared by the host C-ABI layer and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../include/sedef_hip.h"

namespace sdf {

// Scoring constants in the form the kernels consume (bytes of the int8 difference domain).
struct ScoreK {
  int32_t q, e;        // gap open / extend as int8 values
  int32_t qe;          // q + e (int)
  uint8_t q_b;         // (uint8)q
  uint8_t qe2_b;       // (uint8)((q+e)*2)
  uint8_t cap_b;       // (uint8)(mat[0] + (q+e)*2)
  uint8_t sc_match;    // (uint8)mat[0]
  uint8_t sc_mis;      // (uint8)mat[1]
  uint8_t wild;        // m-1
  uint8_t pad_[2];
  int8_t mat[25];      // the whole 5x5 matrix: KSW_EZ_GENERIC_SC scores by mat[target * m + query] (reference :139-141)
  int8_t pad2_[3];
};

// One planned task as the kernels see it.
struct PlanTask {
  int64_t q_word;    // word offset of the packed query in the pool
  int64_t t_word;    // word offset of the packed target
  int64_t dir_off;   // byte offset of this task's direction matrix in the workspace
  int64_t cig_slot;  // word offset of this task's CIGAR staging slot
  int32_t qlen, tlen;
  int32_t w;         // resolved band (never negative)
  int32_t zdrop;
  int32_t flag;
  int32_t ncol16;    // direction-matrix row stride in cells (n_col_*16 of the reference)
  int32_t out_idx;   // index of the result record
  int32_t cig_cap;   // words in the staging slot
  int32_t nreg;      // 0: general kernel, byte-per-cell direction rows; >0: wave kernel with nreg
                     // packed registers, direction flags in 16-row x 128-slot bit blocks
  int32_t pad_;      // 1: general kernel with its state in an HBM slab; 2: pair kernel (extz2_pair.hip): nreg counts
                     // 64-slot registers and the flags are per-task uint2 records; 3 / 4: general kernel, PLAIN
                     // flavour (packed recurrence, H along the band edge only), state in LDS / in an HBM slab
};

// Per-anti-diagonal band geometry (reference: extern/ksw2_extz2_sse.cc:101-115).
struct Band {
  int lo0, hi0;  // logical band [st0, en0]
  int lo, hi;    // widened to whole 16-cell blocks [st, en]
};

__host__ __device__ inline bool band_of(int r, int qlen, int tlen, int w, Band &b) {
  int lo = 0, hi = tlen - 1;
  if (lo < r - qlen + 1) lo = r - qlen + 1;
  if (hi > r) hi = r;
  if (lo < ((r - w + 1) >> 1)) lo = (r - w + 1) >> 1;
  if (hi > ((r + w) >> 1)) hi = (r + w) >> 1;
  b.lo0 = lo;
  b.hi0 = hi;
  b.lo = lo / 16 * 16;            // lo >= 0 whenever lo <= hi
  b.hi = (hi + 16) / 16 * 16 - 1;
  return lo <= hi;
}

#ifdef __HIPCC__
// ---- packed 16-bit helpers of the DP kernels: every state byte of the reference is held as value << 8 in a
// 16-bit half, so the packed ALU reproduces the reference's wrap-around int8 arithmetic two cells at a time ----
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

#define SDF_OPQ(x) asm("" : "+v"(x))  // make a value opaque to instcombine (keeps the packed forms)

__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ unsigned pk_maxi(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, a),
                                                                __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned pk_maxu(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a),
                                                                __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_minu(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a),
                                                                __builtin_bit_cast(u16x2, b)));
}
// max(a - b, 0) per half: one v_pk_sub_u16 with the clamp bit (unsigned saturation)
__device__ __forceinline__ unsigned pk_subsat_u(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// min(x, 1) per half = "x != 0" as 0/1.  Written as the instruction itself: the optimiser would
// otherwise turn it into per-half compares + selects.
__device__ __forceinline__ unsigned pk_nonzero_(unsigned a, unsigned one_opaque) {
  return pk_minu(a, one_opaque);
}
#define pk_nonzero(a) pk_nonzero_((a), one2)
// F <- (F << 1) | bit, as the single instruction it is
__device__ __forceinline__ unsigned shl1_or(unsigned f, unsigned bit) {
  return (f << 1) + bit;  // bit 0 of f << 1 is clear: + == |, and it selects as one v_lshl_add_u32
}
__device__ __forceinline__ unsigned pk_mad(unsigned a, unsigned b, unsigned c) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) +
                                          __builtin_bit_cast(u16x2, c));
}
__device__ __forceinline__ unsigned pk_ashr15(unsigned a) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(i16x2, a) >> (i16x2){15, 15});
}
__device__ __forceinline__ unsigned pk_shl(unsigned a, unsigned n) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, a)
                                          << (u16x2){(unsigned short)n, (unsigned short)n});
}

#endif  // __HIPCC__

}  // namespace sdf
