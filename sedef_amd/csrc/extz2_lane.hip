// Lane kernel: ONE LANE per small full-band task, 64 tasks per wavefront.
//
// What `sedef align` asks the DP for is mostly small: the gap fills between the anchors of a chain are at most 210 x 209
// (reference: src/align.cc:235, src/chain.cc:146-149,157-158) and 59 % of an hg19-shaped task stream is at most 100 cells
// (SURVEY 8d, config 4).  The window kernels give such a task a wavefront (or half of one): a 50 x 50 task fills 40 % of
// the slots of its 64-slot window for a hundred rows, and costs a plan record, a pairing probe and a launch-order entry on
// the host.  Here a task is the private work of one lane: the lane walks its own matrix row by row (query position j
// outside, target position i inside), computing exactly its cells and nothing else; the 64 lanes of a wavefront run in
// lockstep over the largest matrix among them, and the tasks are sorted by (qlen, tlen) on the device so that the
// matrices of a wavefront are (nearly) the same.  Nothing per task is prepared on the host: it marks eligible tasks
// during the scan of the batch and uploads 16 bytes for each (sdf_plan.hip: LaneRec); sorting (hipCUB radix sort),
// CIGAR-slot and flag-region offsets (two scans) and the plan records the traceback reads are made by kernels below.
//
// The recurrence is ksw_extz2_sse's (reference: extern/ksw2_extz2_sse.cc:26-47,172-194) cell for cell, evaluated in
// another order: cell (i, j) takes x, v from (i - 1, j) -- carried in registers along the row -- and u, y from
// (i, j - 1) -- one LDS word per target position and lane, which also holds the target base --, with the reference's
// border values (:117-121: x = 0, v = q below the first row; y = 0, u = q beyond the first column).  Eligible are
// full-band tasks (w < 0 or w >= both lengths: all of SEDEF's own calls) under a "tame" scoring -- match + 2 (q + e) <=
// 127, so that every byte of the reference's state stays in 0..127 and its wrap-around, signed/unsigned and
// sign-extension artefacts (:145-146) cannot occur; with a full band no cell outside the matrix feeds a cell inside it
// ("hard parts").  Everything else keeps to the window kernels.  The exact H is followed along the first
// column and, through the row sums of u, along the last one (score, mte; :226-267 read u, v as bytes).
//
// Direction flags: 4 bits per cell (a > z | b > z' in bits 0-1 as 0 / 1 / 2, x > 0, y > 0), rows of ceil(tlen / 8)
// words per query position, one region per task (traceback.hip: layout 5).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "sdf_internal.h"

namespace sdf {

struct LaneRec {      // what the host uploads per task of the batch (16 bytes; invalid: flag = 0xffff)
  uint32_t q_word, t_word;  // word offsets of the packed sequences in the pool
  uint32_t out_idx;
  uint8_t qlen_m1, tlen_m1;  // lengths - 1 (1 .. 256)
  uint16_t flag;             // SDF_FLAG_SCORE_ONLY | SDF_FLAG_REV_CIGAR
};

constexpr int kLaneMaxLen = 256;      // longest sequence of a lane task
constexpr int kLaneMaxCells = 16384;  // most cells of a lane task: a lane alone on its row costs ~100 cycles per cell

__host__ __device__ inline int lane_row_words(int tlen) { return (tlen + 7) >> 3; }
__host__ __device__ inline size_t lane_dir_bytes(int qlen, int tlen) { return (size_t)qlen * (size_t)lane_row_words(tlen) * 4; }
// launch classes by target length: the LDS of a wavefront is 256 bytes per target position of its longest task
__host__ __device__ inline int lane_class(int tlen) { return tlen <= 32 ? 0 : tlen <= 64 ? 1 : tlen <= 128 ? 2 : 3; }
__host__ __device__ inline size_t lane_lds_bytes(int cls) { return (size_t)256 * (size_t)((32 << cls) + 1); }

__device__ __forceinline__ int lane_wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, 64);
    v = o > v ? o : v;
  }
  return __builtin_amdgcn_readfirstlane(v);
}

// ---- planning on the device ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lane_keys_kernel(const LaneRec *__restrict__ recs, int n, uint32_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const LaneRec r = recs[k];
  // class, query length, target length: the tasks of a wavefront get (nearly) equal matrices; others sort to the end
  keys[k] = r.flag == 0xffffu ? 0xfffffu : ((uint32_t)lane_class(r.tlen_m1 + 1) << 16) | ((uint32_t)r.qlen_m1 << 8) | r.tlen_m1;
  vals[k] = (uint32_t)k;
}

// per sorted position: CIGAR staging words and direction-flag bytes (scanned into offsets)
__global__ __launch_bounds__(256) void lane_sizes_kernel(const LaneRec *__restrict__ recs, const uint32_t *__restrict__ vals,
                                                         int n_lane, unsigned long long *__restrict__ cap,
                                                         unsigned long long *__restrict__ dirb) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_lane) return;
  const LaneRec r = recs[vals[p]];
  const int ql = r.qlen_m1 + 1, tl = r.tlen_m1 + 1;
  const bool with_dir = !(r.flag & SDF_FLAG_SCORE_ONLY);
  cap[p] = with_dir ? (unsigned long long)(ql + tl + 2) : 0ull;
  dirb[p] = with_dir ? (unsigned long long)lane_dir_bytes(ql, tl) : 0ull;
}

__global__ __launch_bounds__(256) void lane_plan_kernel(const LaneRec *__restrict__ recs, const uint32_t *__restrict__ vals,
                                                        int n_lane, const unsigned long long *__restrict__ cap_off,
                                                        const unsigned long long *__restrict__ dir_off, int64_t stage0,
                                                        int64_t dir0, PlanTask *__restrict__ plan) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_lane) return;
  const LaneRec r = recs[vals[p]];
  PlanTask t;
  t.q_word = r.q_word;
  t.t_word = r.t_word;
  t.dir_off = dir0 + (int64_t)dir_off[p];
  t.cig_slot = stage0 + (int64_t)cap_off[p];
  t.qlen = r.qlen_m1 + 1;
  t.tlen = r.tlen_m1 + 1;
  t.w = t.qlen > t.tlen ? t.qlen : t.tlen;
  t.zdrop = -1;
  t.flag = r.flag;
  t.ncol16 = ((t.qlen < t.tlen ? t.qlen : t.tlen) + 15) / 16 * 16 + 16;
  t.out_idx = (int32_t)r.out_idx;
  t.cig_cap = (r.flag & SDF_FLAG_SCORE_ONLY) ? 0 : t.qlen + t.tlen + 2;
  t.nreg = 1;
  t.pad_ = 8;  // direction-flag layout 5 (traceback.hip)
  plan[p] = t;
}

// ---- the DP -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void extz2_lane_kernel(const PlanTask *__restrict__ plan, int n,
                                                        const uint32_t *__restrict__ pool, ScoreK sc,
                                                        uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint32_t lane_lds[];  // [target position][lane]: u | y << 8 | target base << 16
  const int lane = threadIdx.x;
  const int p = blockIdx.x * 64 + lane;
  const bool have = p < n;
  const PlanTask tk = plan[have ? p : n - 1];
  const int qlen = have ? tk.qlen : 0, tlen = have ? tk.tlen : 0;
  const int Qw = lane_wave_max(qlen), Tw = lane_wave_max(tlen);
  const uint32_t *tw = pool + tk.t_word, *tn = tw + (tk.tlen + 15) / 16;
  const uint32_t *qw = pool + tk.q_word, *qn = qw + (tk.qlen + 15) / 16;
  const int gq = sc.q, qe = sc.qe;
  uint32_t *col = lane_lds + lane;

  // first row's upper neighbours (:121: y = 0, u = q beyond the first column) and the target bases (N: 4)
  {
    uint32_t cw = 0, nm = 0;
    for (int i = 0; i < Tw; ++i) {
      if (i < tlen) {
        if ((i & 15) == 0) cw = tw[i >> 4];
        if ((i & 31) == 0) nm = tn[i >> 5];
        const uint32_t c = ((nm >> (i & 31)) & 1u) ? 4u : ((cw >> ((i & 15) * 2)) & 3u);
        col[i * 64] = (i ? (uint32_t)gq : 0u) | (c << 16);
      }
    }
  }
  const int zm = (int)(int8_t)sc.sc_match + 2 * qe, zmis = (int)(int8_t)sc.sc_mis + 2 * qe, zwild = 2 * qe;
  const int cap = (int)(int8_t)sc.sc_match + 2 * qe;
  const bool with_dir = have && !(tk.flag & SDF_FLAG_SCORE_ONLY);
  const int rw = lane_row_words(tlen);
  uint32_t *dirp = reinterpret_cast<uint32_t *>(dirbase + tk.dir_off);

  int32_t h0 = 0;                // exact H of cell (0, j)
  int32_t mte = SDF_NEG_INF, mte_j = -1, score = SDF_NEG_INF;
  uint32_t qcw = 0, qnm = 0;
  for (int j = 0; j < Qw; ++j) {
    const bool row_on = j < qlen;
    if (row_on) {
      if ((j & 15) == 0) qcw = qw[j >> 4];
      if ((j & 31) == 0) qnm = qn[j >> 5];
    }
    const bool q_n = ((qnm >> (j & 31)) & 1u) != 0u;
    const uint32_t qc = (qcw >> ((j & 15) * 2)) & 3u;
    const int z_eq = q_n ? zwild : zm, z_ne = q_n ? zwild : zmis;
    int x = 0, v = j ? gq : 0;  // left of the first column (:120)
    int usum = 0;
    uint32_t dw = 0u;
    uint32_t w = col[0];
    for (int i = 0; i < Tw; ++i) {
      const uint32_t wn = col[(i + 1) * 64];  // (the next position's word: one row of slack behind the last)
      if (row_on && i < tlen) {
        const int uo = (int)(w & 0xffu), yo = (int)((w >> 8) & 0xffu);
        const uint32_t tc = w >> 16;
        int z = tc == qc ? z_eq : z_ne;
        z = tc == 4u ? zwild : z;
        int a = x + v, b = yo + uo;
        const uint32_t fa = a > z ? 1u : 0u;  // ties: diagonal before E before F (:173-178)
        z = a > z ? a : z;
        const uint32_t d = b > z ? 2u : fa;
        z = b > z ? b : z;
        z = z < cap ? z : cap;
        const int un = z - v, vn = z - uo;
        z -= gq;
        a -= z;
        b -= z;
        x = a > 0 ? a : 0;
        const int yn = b > 0 ? b : 0;
        const uint32_t nib = d | (a > 0 ? 4u : 0u) | (b > 0 ? 8u : 0u);
        dw = (dw >> 4) | (nib << 28);
        v = vn;
        col[i * 64] = (uint32_t)un | ((uint32_t)yn << 8) | (tc << 16);
        if (i == 0) h0 += j ? vn - qe : vn - 2 * qe;  // (:249: H(0,0) = v - 2 (q + e); :231 along the first column)
        else usum += un;
        if ((i & 7) == 7 && with_dir) dirp[(size_t)j * rw + (i >> 3)] = dw;
      }
      w = wn;
    }
    if (row_on) {
      if (with_dir && (tlen & 7)) dirp[(size_t)j * rw + (tlen >> 3)] = dw >> ((8 - (tlen & 7)) * 4);
      // exact H of the row's last cell: along the first column to (0, j), then along the row (:231, u read as a byte)
      const int32_t hl = h0 + usum - (tlen - 1) * qe;
      if (hl > mte) {  // (:252: strict, rows in ascending order)
        mte = hl;
        mte_j = j;
      }
      score = hl;  // (the last row's value stays: :255)
    }
  }
  if (have) {
    sdf_result o;
    o.score = score;
    o.max = 0;
    o.max_q = o.max_t = -1;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = mte;
    o.mte_q = mte_j + (tlen - 1) - ((tlen - 1) | 15);  // (r - en with the reference's block-rounded en, :262)
    o.zdropped = 0;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[tk.out_idx] = o;
  }
}

}  // namespace sdf
