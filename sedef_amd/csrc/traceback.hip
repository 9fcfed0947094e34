// Traceback (direction matrix -> CIGAR + column counts) and CIGAR compaction for gfx950.
//
// Restates ksw_backtrack / ksw_push_cigar (reference: extern/ksw2.h:98-151, rotated layout)
// and the counters of populate_nice_alignment (reference: src/align.cc:274-315).
//
// The walk is inherently serial per task (each step's address depends on the previous
// direction byte), so it is parallelised across tasks: one lane per task, every lane chasing
// its own path.  Runs are emitted from the END of the task's staging slot towards its start,
// which leaves them in forward order without a reversal pass.
#include <hip/hip_runtime.h>

#include "sdf_internal.h"

namespace sdf {

__device__ __forceinline__ uint32_t code_at(const uint32_t *codes, const uint32_t *nmask, int k) {
  const uint32_t c = (codes[k >> 4] >> ((k & 15) * 2)) & 3u;
  const uint32_t n = (nmask[k >> 5] >> (k & 31)) & 1u;
  return n ? 4u : c;
}

// LAYOUT: 0 = byte rows (general kernel), 1 = wave-kernel bit blocks, 2 = pair-kernel bit blocks, 3 = wave-kernel
// bit blocks per target stripe (extz2_stripe.hip).  One
// instantiation per layout: in a common loop every step would wait for all outstanding loads at the point where
// the three fetch paths meet.  Tasks of another layout leave at once (the host launches only the instantiations
// a chunk needs).
template <int LAYOUT>
__global__ __launch_bounds__(64) void traceback_kernel(const PlanTask *__restrict__ plan, int n,
                                                       const uint32_t *__restrict__ pool,
                                                       const uint8_t *__restrict__ dirbase,
                                                       sdf_result *__restrict__ res,
                                                       uint32_t *__restrict__ stage) {
  // n < 0: -n tasks, ONE per wavefront (lane 0 walks): the lanes of a wavefront leave their cached lines at
  // different steps, so with 64 walks per wavefront every step waits for somebody's miss; a chunk of few, long
  // tasks is walked faster with a wavefront each
  const bool solo = n < 0;
  if (solo && threadIdx.x != 0) return;
  const int k = solo ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= (solo ? -n : n)) return;
  const PlanTask tk = plan[k];
  if ((tk.nreg == 0 ? 0 : tk.pad_ == 2 ? 2 : tk.pad_ == 5 ? 3 : 1) != LAYOUT) return;
  sdf_result rr = res[tk.out_idx];
  if (tk.flag & SDF_FLAG_SCORE_ONLY) return;

  int i, j;  // (sequences are shorter than 2^31)
  if (!rr.zdropped && !(tk.flag & SDF_FLAG_EXTZ_ONLY)) {
    i = tk.tlen - 1;
    j = tk.qlen - 1;
  } else if (rr.max_t >= 0 && rr.max_q >= 0) {
    i = rr.max_t;
    j = rr.max_q;
  } else {
    return;  // n_cigar stays 0
  }

  const uint8_t *dir = dirbase + tk.dir_off;
  const int64_t stride = tk.ncol16;
  const uint32_t *tw = pool + tk.t_word, *tn = tw + (tk.tlen + 15) / 16;
  const uint32_t *qw = pool + tk.q_word, *qn = qw + (tk.qlen + 15) / 16;
  uint32_t *slot = stage + tk.cig_slot;
  int pos = tk.cig_cap;
  int cur_op = -1, cur_len = 0;
  int32_t matches = 0, mismatches = 0, gaps = 0, gap_bases = 0;

  auto push = [&](int op, int len) {
    if (op == cur_op) {
      cur_len += len;
    } else {
      if (cur_op >= 0) slot[--pos] = ((uint32_t)cur_len << 4) | (uint32_t)cur_op;
      cur_op = op;
      cur_len = len;
    }
  };

  // The two sequences are read backwards, one base per step that consumes them: the current 16-base code word and
  // 32-base N-mask word of each are kept SHIFTED so that the base under the walk is in the top bits -- a step is one
  // shift, a new word is loaded every 16 steps.
  uint32_t qsh = 0, qnsh = 0, tsh = 0, tnsh = 0;
  if (i >= 0 && j >= 0) {
    qsh = qw[j >> 4] << ((15 - (j & 15)) * 2);
    qnsh = qn[j >> 5] << (31 - (j & 31));
    tsh = tw[i >> 4] << ((15 - (i & 15)) * 2);
    tnsh = tn[i >> 5] << (31 - (i & 31));
  }
  int state = 0;
  // wave-kernel layout: block rb = r/16 holds, per packed register k and lane, one uint4 of four
  // 32-bit flag words (a>z, b>z', x>0, y>0); bit 15-(r%16) (+16 for the odd slot) is row r;
  // slot = t - (band start of row 16*rb).  The last fetched uint4 is kept: a path stays inside
  // one 16-row x 2-slot tile for several steps.
  const uint4 *dirw = reinterpret_cast<const uint4 *>(dir);
  int cached_line = -1;  // index/4 of the 64-byte line (4 lanes = 8 slots x 16 rows) held below
  uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0;
  int blk_rb = -1, blk_sb = -1, blk_base = 0;
  while ((i | j) >= 0) {
    const int r = i + j;
    Band b;
    band_of(r, tk.qlen, tk.tlen, tk.w, b);
    int forced = -1;
    uint32_t d = 0;
    if (i < b.lo) forced = 2;
    if (i > b.hi) forced = 1;
    if (forced < 0) {
      if (LAYOUT == 0) {
        d = dir[(int64_t)r * stride + (i - b.lo)];
      } else if (LAYOUT == 2) {
        // pair-kernel layout: per register k (64 slots) and lane one uint2 (a | b << 16, x | y << 16) of
        // 16-bit row masks; a 64-byte line holds 8 slots x 16 rows
        const int rb = r >> 4;
        if (rb != blk_rb) {
          Band b0;
          band_of(rb << 4, tk.qlen, tk.tlen, tk.w, b0);
          blk_rb = rb;
          blk_base = b0.lo;
        }
        const int idx = rb * (tk.nreg * 64) + (i - blk_base);  // (register k, lane l) = slot 64k + l
        if ((idx >> 3) != cached_line) {
          cached_line = idx >> 3;
          const uint4 *ln = dirw + ((int64_t)cached_line << 2);
          c0 = ln[0];
          c1 = ln[1];
          c2 = ln[2];
          c3 = ln[3];
        }
        const int sel = (idx & 7) >> 1;
        const uint4 cached = sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
        const uint32_t wab = (idx & 1) ? cached.z : cached.x, wxy = (idx & 1) ? cached.w : cached.y;
        const int bit = 15 - (r & 15);
        const uint32_t fa = (wab >> bit) & 1u, fb = (wab >> (bit + 16)) & 1u;
        const uint32_t fx = (wxy >> bit) & 1u, fy = (wxy >> (bit + 16)) & 1u;
        d = (fb ? 2u : fa) | (fx << 3) | (fy << 4);
      } else if (LAYOUT == 3) {
        // stripes of 128 * nreg target positions, each a stand-alone wave-kernel task over its slice in local
        // coordinates (row r - T0, position t - T0); one flag region per stripe
        const int sw = 128 * tk.nreg;
        const int sb = i / sw, t0 = sb * sw;
        const int rp = r - t0, tp = i - t0;
        const int rb = rp >> 4;
        if (rb != blk_rb || sb != blk_sb) {
          Band b0;
          band_of(rb << 4, tk.qlen, tk.tlen - t0 < sw ? tk.tlen - t0 : sw, tk.w, b0);
          blk_rb = rb;
          blk_sb = sb;
          blk_base = b0.lo;
        }
        const int slot = tp - blk_base;
        const int per_stripe = ((tk.qlen + sw - 1 + 15) / 16) * tk.nreg * 64;  // uint4 records
        const int idx = sb * per_stripe + rb * (tk.nreg * 64) + (slot >> 1);  // (register k, lane l) = slots 128k + 2l, +1
        if ((idx >> 2) != cached_line) {
          cached_line = idx >> 2;
          const uint4 *ln = dirw + ((int64_t)cached_line << 2);
          c0 = ln[0];
          c1 = ln[1];
          c2 = ln[2];
          c3 = ln[3];
        }
        const int sel = idx & 3;
        const uint4 cached = sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
        const int bit = 15 - (rp & 15) + ((slot & 1) << 4);
        const uint32_t fa = (cached.x >> bit) & 1u, fb = (cached.y >> bit) & 1u;
        const uint32_t fx = (cached.z >> bit) & 1u, fy = (cached.w >> bit) & 1u;
        d = (fb ? 2u : fa) | (fx << 3) | (fy << 4);
      } else {
        const int rb = r >> 4;
        if (rb != blk_rb) {
          Band b0;
          band_of(rb << 4, tk.qlen, tk.tlen, tk.w, b0);
          blk_rb = rb;
          blk_base = b0.lo;
        }
        const int slot = i - blk_base;
        const int idx = rb * (tk.nreg * 64) + (slot >> 1);
        if ((idx >> 2) != cached_line) {  // a diagonal run stays inside one line for ~8 steps
          cached_line = idx >> 2;
          const uint4 *ln = dirw + ((int64_t)cached_line << 2);
          c0 = ln[0];
          c1 = ln[1];
          c2 = ln[2];
          c3 = ln[3];
        }
        const int sel = idx & 3;
        const uint4 cached = sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
        const int bit = 15 - (r & 15) + ((slot & 1) << 4);
        const uint32_t fa = (cached.x >> bit) & 1u, fb = (cached.y >> bit) & 1u;
        const uint32_t fx = (cached.z >> bit) & 1u, fy = (cached.w >> bit) & 1u;
        d = (fb ? 2u : fa) | (fx << 3) | (fy << 4);
      }
    }
    if (state == 0) state = (int)(d & 7u);
    else if (!((d >> (state + 2)) & 1u)) state = 0;
    if (state == 0) state = (int)(d & 7u);
    if (forced >= 0) state = forced;
    // one step: state 0 consumes a base of both (M), 1 / 3 of the target (ksw D, op 2), anything else of the query
    const bool step_i = state == 0 || state == 1 || state == 3, step_j = state == 0 || !step_i;
    if (state == 0) {
      // match <=> neither base is N and the 2-bit codes agree (src/align.cc:29-35 on the codes)
      if ((((qnsh | tnsh) >> 31) | ((qsh ^ tsh) >> 30)) == 0u) ++matches; else ++mismatches;
    }
    push(state == 0 ? 0 : step_i ? 2 : 1, 1);
    if (step_i) {
      --i;
      if ((i & 15) == 15) {  // (also true for i == -1: the loop ends before the words are used)
        if (i >= 0) {
          tsh = tw[i >> 4];
          tnsh = (i & 31) == 31 ? tn[i >> 5] : tnsh << 1;
        }
      } else {
        tsh <<= 2;
        tnsh <<= 1;
      }
    }
    if (step_j) {
      --j;
      if ((j & 15) == 15) {
        if (j >= 0) {
          qsh = qw[j >> 4];
          qnsh = (j & 31) == 31 ? qn[j >> 5] : qnsh << 1;
        }
      } else {
        qsh <<= 2;
        qnsh <<= 1;
      }
    }
  }
  if (i >= 0) push(2, i + 1);
  if (j >= 0) push(1, j + 1);
  if (cur_op >= 0) slot[--pos] = ((uint32_t)cur_len << 4) | (uint32_t)cur_op;

  const int ncig = tk.cig_cap - pos;
  for (int c = pos; c < tk.cig_cap; ++c) {
    const uint32_t wd = slot[c];
    if (wd & 0xfu) {
      ++gaps;
      gap_bases += (int32_t)(wd >> 4);
    }
  }
  rr.n_cigar = ncig;
  rr.matches = matches;
  rr.mismatches = mismatches;
  rr.gaps = gaps;
  rr.gap_bases = gap_bases;
  res[tk.out_idx] = rr;
}

template __global__ void traceback_kernel<0>(const PlanTask *, int, const uint32_t *, const uint8_t *, sdf_result *,
                                             uint32_t *);
template __global__ void traceback_kernel<1>(const PlanTask *, int, const uint32_t *, const uint8_t *, sdf_result *,
                                             uint32_t *);
template __global__ void traceback_kernel<2>(const PlanTask *, int, const uint32_t *, const uint8_t *, sdf_result *,
                                             uint32_t *);
template __global__ void traceback_kernel<3>(const PlanTask *, int, const uint32_t *, const uint8_t *, sdf_result *,
                                             uint32_t *);

// Exclusive scan of n_cigar over the result records in record order -> cigar_off, in three small launches:
// per 1024-record block a local scan + the block total, a scan of the block totals (one workgroup), and the
// addition of the block offsets.  `part` holds ceil(n / 1024) + 1 words; the grand total goes to *total.
__global__ __launch_bounds__(1024) void cigar_scan_blocks_kernel(sdf_result *__restrict__ res, int n,
                                                                 unsigned long long *__restrict__ part) {
  __shared__ unsigned long long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = blockIdx.x * 1024 + tid;
  const unsigned long long mine = k < n ? (unsigned long long)res[k].n_cigar : 0ull;
  unsigned long long inc = mine;  // inclusive scan inside the wavefront
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(inc, off);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  unsigned long long before = 0;
  for (int q = 0; q < wv; ++q) before += wsum[q];
  if (k < n) res[k].cigar_off = (int64_t)(before + inc - mine);
  if (tid == 1023) part[blockIdx.x] = before + inc;
}

__global__ __launch_bounds__(1024) void cigar_scan_parts_kernel(unsigned long long *__restrict__ part, int nb,
                                                                unsigned long long *__restrict__ total) {
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int k = base + tid;
    const unsigned long long mine = k < nb ? part[k] : 0ull;
    unsigned long long inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long up = __shfl_up(inc, off);
      if (lane >= off) inc += up;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned long long before = carry_s;
    for (int q = 0; q < wv; ++q) before += wsum[q];
    if (k < nb) part[k] = before + inc - mine;
    __syncthreads();
    if (tid == 1023) carry_s = before + inc;
    __syncthreads();
  }
  if (tid == 0) *total = carry_s;
}

__global__ __launch_bounds__(256) void cigar_scan_add_kernel(sdf_result *__restrict__ res, int n,
                                                             const unsigned long long *__restrict__ part) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) res[k].cigar_off += (int64_t)part[k >> 10];
}

// Copy every task's CIGAR from its staging slot to its compact position.  One wave per task.
__global__ __launch_bounds__(256) void cigar_compact_kernel(const PlanTask *__restrict__ plan, int n,
                                                            const sdf_result *__restrict__ res,
                                                            const uint32_t *__restrict__ stage,
                                                            uint32_t *__restrict__ out,
                                                            unsigned long long cap) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const int lane = threadIdx.x & 63;
  const PlanTask tk = plan[k];
  const sdf_result rr = res[tk.out_idx];
  const int nc = rr.n_cigar;
  if ((unsigned long long)rr.cigar_off + (unsigned long long)nc > cap) return;
  const uint32_t *src = stage + tk.cig_slot + (tk.cig_cap - nc);
  uint32_t *dst = out + rr.cigar_off;
  if (tk.flag & SDF_FLAG_REV_CIGAR) {
    for (int c = lane; c < nc; c += 64) dst[c] = src[nc - 1 - c];
  } else {
    for (int c = lane; c < nc; c += 64) dst[c] = src[c];
  }
}

}  // namespace sdf
