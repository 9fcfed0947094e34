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
                     // 64-slot registers and the flags are per-task uint2 records
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

}  // namespace sdf
