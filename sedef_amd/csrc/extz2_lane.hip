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
// Direction flags: 4 bits per cell (a > z | b > z' in bits 0-1 as 0 / 1 / 2, x > 0, y > 0), 8 bytes per query position
// and tile of 16 target positions, tile after tile, one region per task (traceback.hip: layout 5).
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

// direction flags of a task: per column tile of 16 target positions one 8-byte record per query position (four 16-bit flag planes: a bit per
// cell), the records of a tile back to back -- a lane writes its region front to back, 8 bytes per row of a tile
__host__ __device__ inline size_t lane_dir_bytes(int qlen, int tlen) { return (size_t)((tlen + 15) >> 4) * (size_t)qlen * 8; }
// launch classes by query length: the LDS of a wavefront is 128 bytes per query position of its longest task
__host__ __device__ inline int lane_class(int qlen) { return qlen <= 32 ? 0 : qlen <= 64 ? 1 : qlen <= 128 ? 2 : 3; }
__host__ __device__ inline size_t lane_lds_bytes(int cls) { return (size_t)128 * (size_t)((32 << cls) + 2); }

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
  keys[k] = r.flag == 0xffffu ? 0xfffffu : ((uint32_t)lane_class(r.qlen_m1 + 1) << 16) | ((uint32_t)r.qlen_m1 << 8) | r.tlen_m1;
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

// ---- planning on the device, second form (round 5): counting sort, no workgroup waits for another ------------------
// hipCUB's radix sort and scans pass their partial results from workgroup to workgroup (decoupled look-back): next to
// kernels that hold every wavefront slot for milliseconds -- the chains and strips of the long tasks, launched first -- a
// workgroup that has a slot waits for one that has none, and the sort of a million keys took 4.5-4.9 ms instead of 0.4
// (profiles/r05_hg19_timeline_before.txt) with the lane DP, 38 % of the batch's work, queued behind it.  The key has 19 bits
// -- class, score-only, query length, target length -- and every task of a key needs the same CIGAR slot and flag region:
// a histogram, one scan over the BINS (count, staging words, flag bytes: three sums at once) and a pass that gives every
// task its rank in its bin by an atomic add and writes its plan record.  Tasks of a bin are interchangeable (same matrix
// size), so their order inside the bin does not matter to anything but the layout of the workspace.
constexpr int kLaneKeyBits = 19, kLaneBins = 1 << kLaneKeyBits, kLaneScanBlock = 1024;
__device__ __forceinline__ uint32_t lane_key(const LaneRec &r) {
  return ((uint32_t)lane_class(r.qlen_m1 + 1) << 17) | ((r.flag & SDF_FLAG_SCORE_ONLY) ? 1u << 16 : 0u) | ((uint32_t)r.qlen_m1 << 8) | r.tlen_m1;
}
__device__ __forceinline__ uint32_t lane_key_cap(uint32_t key) {  // CIGAR staging words of a task of this key
  return (key & (1u << 16)) ? 0u : ((key >> 8) & 0xffu) + (key & 0xffu) + 2u + 2u;
}
__device__ __forceinline__ uint32_t lane_key_dir(uint32_t key) {  // direction-flag bytes
  return (key & (1u << 16)) ? 0u : (uint32_t)lane_dir_bytes((int)((key >> 8) & 0xffu) + 1, (int)(key & 0xffu) + 1);
}
// One atomic add per KEY a wavefront holds instead of one per task: SEDEF's gap fills come by the hundred thousand with a
// handful of sizes (708,600 tasks of ~25 x ~25 bases in the first round of a chr1-sized bucket), and 64 lanes adding to one
// counter are 64 atomics in a row at the L2 -- the histogram and the placement of that round took 1.9 ms EACH, more than
// its DP (profiles/r06_stage_kernel_timeline.txt); a batch of many sizes (the hg19 mixture: 0.3 ms each) gives up after
// eight keys and takes the rest one lane at a time as before.  Returns the lane's rank in its key's bin (the value its own
// atomic add would have returned) where `rank` is asked for.
template <bool RANK>
__device__ __forceinline__ uint32_t lane_bin_add(uint32_t *__restrict__ bins, const uint32_t key, const bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
  uint32_t mine = 0u;
  for (int round = 0; todo; ++round) {
    if (round == 8) {  // (many keys in this wavefront: the rest lane by lane)
      if ((todo >> lane) & 1ull) mine = atomicAdd(&bins[key], 1u);
      break;
    }
    const int leader = __ffsll((long long)todo) - 1;  // wave-uniform
    const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
    const unsigned long long same = __ballot(active && key == k0) & todo;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(&bins[k0], (uint32_t)__popcll(same));
    if (RANK) {
      base = (uint32_t)__shfl((int)base, leader);
      if ((same >> lane) & 1ull) mine = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    }
    todo &= ~same;
  }
  return mine;
}
__global__ __launch_bounds__(256) void lane_hist_kernel(const LaneRec *__restrict__ recs, int n, uint32_t *__restrict__ count) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const bool have = k < n;  // (whole wavefronts take part in the ballots)
  const LaneRec r = recs[have ? k : n - 1];
  (void)lane_bin_add<false>(count, lane_key(r), have && r.flag != 0xffffu);
}
// per block of 1024 bins: exclusive prefixes inside the block (tasks, staging words, flag bytes) and the block's totals
__global__ __launch_bounds__(256) void lane_bins_scan_kernel(const uint32_t *__restrict__ count, uint32_t *__restrict__ base,
                                                             unsigned long long *__restrict__ capbase,
                                                             unsigned long long *__restrict__ dirbase, uint32_t *__restrict__ tot_cnt,
                                                             unsigned long long *__restrict__ tot_cap,
                                                             unsigned long long *__restrict__ tot_dir) {
  __shared__ unsigned long long sh[3][256];
  const int t = threadIdx.x;
  const uint32_t key0 = (uint32_t)blockIdx.x * kLaneScanBlock + 4u * t;
  uint32_t c[4];
  unsigned long long cap[4], dir[4], sc = 0, sp = 0, sd = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = count[key0 + j];
    cap[j] = (unsigned long long)c[j] * lane_key_cap(key0 + j);
    dir[j] = (unsigned long long)c[j] * lane_key_dir(key0 + j);
    sc += c[j], sp += cap[j], sd += dir[j];
  }
  sh[0][t] = sc, sh[1][t] = sp, sh[2][t] = sd;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // (Hillis-Steele over the 256 partial sums)
    unsigned long long a0 = 0, a1 = 0, a2 = 0;
    if (t >= off) a0 = sh[0][t - off], a1 = sh[1][t - off], a2 = sh[2][t - off];
    __syncthreads();
    sh[0][t] += a0, sh[1][t] += a1, sh[2][t] += a2;
    __syncthreads();
  }
  unsigned long long ec = sh[0][t] - sc, ep = sh[1][t] - sp, ed = sh[2][t] - sd;  // exclusive
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    base[key0 + j] = (uint32_t)ec;
    capbase[key0 + j] = ep;
    dirbase[key0 + j] = ed;
    ec += c[j], ep += cap[j], ed += dir[j];
  }
  if (t == 255) tot_cnt[blockIdx.x] = (uint32_t)sh[0][255], tot_cap[blockIdx.x] = sh[1][255], tot_dir[blockIdx.x] = sh[2][255];
}
// the block totals (kLaneBins / kLaneScanBlock = 512 of them) to exclusive prefixes, in place: one workgroup
__global__ __launch_bounds__(512) void lane_bins_top_kernel(uint32_t *__restrict__ tot_cnt, unsigned long long *__restrict__ tot_cap,
                                                            unsigned long long *__restrict__ tot_dir) {
  __shared__ unsigned long long sh[3][512];
  const int t = threadIdx.x;
  const unsigned long long c = tot_cnt[t], p = tot_cap[t], d = tot_dir[t];
  sh[0][t] = c, sh[1][t] = p, sh[2][t] = d;
  __syncthreads();
  for (int off = 1; off < 512; off <<= 1) {
    unsigned long long a0 = 0, a1 = 0, a2 = 0;
    if (t >= off) a0 = sh[0][t - off], a1 = sh[1][t - off], a2 = sh[2][t - off];
    __syncthreads();
    sh[0][t] += a0, sh[1][t] += a1, sh[2][t] += a2;
    __syncthreads();
  }
  tot_cnt[t] = (uint32_t)(sh[0][t] - c);
  tot_cap[t] = sh[1][t] - p;
  tot_dir[t] = sh[2][t] - d;
}
// every lane task to its place: rank in its bin by an atomic add, the plan record the DP and the traceback read
__global__ __launch_bounds__(256) void lane_place_kernel(const LaneRec *__restrict__ recs, int n, uint32_t *__restrict__ cursor,
                                                         const uint32_t *__restrict__ base, const unsigned long long *__restrict__ capbase,
                                                         const unsigned long long *__restrict__ dirbase, const uint32_t *__restrict__ tot_cnt,
                                                         const unsigned long long *__restrict__ tot_cap,
                                                         const unsigned long long *__restrict__ tot_dir, int64_t stage0, int64_t dir0,
                                                         PlanTask *__restrict__ plan) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const bool have = k < n;
  const LaneRec r = recs[have ? k : n - 1];
  const bool live = have && r.flag != 0xffffu;
  const uint32_t key = lane_key(r), blk = key / kLaneScanBlock;
  const uint32_t rank = lane_bin_add<true>(cursor, key, live);  // (tasks of a bin are interchangeable: any rank will do)
  if (!live) return;
  const size_t p = (size_t)tot_cnt[blk] + base[key] + rank;
  PlanTask t;
  t.q_word = r.q_word;
  t.t_word = r.t_word;
  t.dir_off = dir0 + (int64_t)(tot_dir[blk] + dirbase[key] + (unsigned long long)rank * lane_key_dir(key));
  t.cig_slot = stage0 + (int64_t)(tot_cap[blk] + capbase[key] + (unsigned long long)rank * lane_key_cap(key));
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
// The matrix of a lane is walked in column TILES of 16 target positions: the tile's column state -- u | y << 8 | target
// base << 16 of the cell above, one register per column -- stays in registers for all rows (the row loop is outside, the
// sixteen cells of a row are unrolled: static register indices), x and v run along the row in two more, and what crosses
// a tile's right edge -- x and v of its last column, per row -- waits in LDS for the next tile (16 bits per row and lane,
// fetched one row ahead).  No LDS access and no lane mask inside a full tile's row: a wavefront's tasks have (nearly)
// equal lengths, only the last tile of a row is ragged.
constexpr int kLaneTile = 16;

// Round 6: the four direction flags of a cell are SIGN BITS of differences the recurrence has anyway -- z - a < 0 (a > z),
// max(z, a) - b < 0, (z' - q) - a < 0 (x > 0), (z' - q) - b < 0 (y > 0) -- and each is shifted into an accumulator of its own
// by one v_alignbit (acc << 1 | sign): a subtract and an align per flag where a compare, a select, an or and a shift-or per
// flag built a nibble (~42 -> ~26 of a cell's ~116 cycles).  Column k of the tile ends at bit 15 - k of every accumulator
// (a ragged tile's row is shifted up by what it lacks); the record is the pair kernels' (a | b << 16, x | y << 16).
template <bool RAGGED>
__device__ __forceinline__ void lane_row(uint32_t (&W)[kLaneTile], int &x, int &v, uint32_t &Fa, uint32_t &Fb, uint32_t &Fx,
                                         uint32_t &Fy, const int ncol, const uint32_t qc, const int z_eq, const int z_ne,
                                         const int zwild, const int cap, const int gq) {
#pragma unroll
  for (int k = 0; k < kLaneTile; ++k) {
    if (!RAGGED || k < ncol) {
      const uint32_t w = W[k];
      const int uo = (int)(w & 0xffu), yo = (int)((w >> 8) & 0xffu);
      const uint32_t tc = w >> 16;
      int z = tc == qc ? z_eq : z_ne;
      z = tc == 4u ? zwild : z;
      const int a = x + v, b = yo + uo;
      Fa = __builtin_amdgcn_alignbit(Fa, (uint32_t)(z - a), 31);  // ties: diagonal before E before F (:173-178)
      z = a > z ? a : z;
      Fb = __builtin_amdgcn_alignbit(Fb, (uint32_t)(z - b), 31);
      z = b > z ? b : z;
      z = z < cap ? z : cap;
      const int un = z - v, vn = z - uo;
      z -= gq;
      const int na = z - a, nb = z - b;  // (negative: the gap goes on)
      Fx = __builtin_amdgcn_alignbit(Fx, (uint32_t)na, 31);
      Fy = __builtin_amdgcn_alignbit(Fy, (uint32_t)nb, 31);
      x = na < 0 ? -na : 0;
      const int yn = nb < 0 ? -nb : 0;
      v = vn;
      W[k] = (uint32_t)un | ((uint32_t)yn << 8) | (w & 0xffff0000u);
    }
  }
  if (RAGGED) {
    const int up = kLaneTile - ncol;
    Fa <<= up, Fb <<= up, Fx <<= up, Fy <<= up;
  }
}

__global__ __launch_bounds__(64) void extz2_lane_kernel(const PlanTask *__restrict__ plan, int n,
                                                        const uint32_t *__restrict__ pool, ScoreK sc,
                                                        uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint16_t lane_edge[];  // [query position][lane]: x | v << 8 at a tile's right edge
  const int lane = threadIdx.x;
  // (the tasks are sorted by size, ascending: the largest go first -- a launch ends with the wavefront that starts last)
  const int p = ((int)gridDim.x - 1 - (int)blockIdx.x) * 64 + lane;
  const bool have = p < n;
  const PlanTask tk = plan[have ? p : n - 1];
  const int qlen = have ? tk.qlen : 0, tlen = have ? tk.tlen : 0;
  const int Qw = lane_wave_max(qlen), Tw = lane_wave_max(tlen);
  const int Tmin = -lane_wave_max(have ? -tlen : -kLaneMaxLen);  // tiles below it are full for every lane that has a task
  const uint32_t *tw = pool + tk.t_word, *tn = tw + (tk.tlen + 15) / 16;
  const uint32_t *qw = pool + tk.q_word, *qn = qw + (tk.qlen + 15) / 16;
  const int gq = sc.q, qe = sc.qe;
  uint16_t *edge = lane_edge + lane;
  const int zm = (int)(int8_t)sc.sc_match + 2 * qe, zmis = (int)(int8_t)sc.sc_mis + 2 * qe, zwild = 2 * qe;
  const int cap = zm;
  const bool with_dir = have && !(tk.flag & SDF_FLAG_SCORE_ONLY);
  uint2 *dirp = reinterpret_cast<uint2 *>(dirbase + tk.dir_off);

  int32_t hrow0 = 0;  // exact H of the last cell of row 0 seen so far: (t0 + ncol - 1, 0)
  int32_t hl = 0, mte = SDF_NEG_INF, mte_j = -1, score = SDF_NEG_INF;
  for (int t0 = 0; t0 < Tw; t0 += kLaneTile) {
    const bool tile_on = t0 < tlen;                 // this lane has columns in the tile
    const int ncol = tile_on ? (tlen - t0 < kLaneTile ? tlen - t0 : kLaneTile) : 0;
    const bool full = t0 + kLaneTile <= Tmin;       // wave-uniform
    const bool last_tile = tile_on && tlen <= t0 + kLaneTile;
    const bool more = t0 + kLaneTile < Tw;          // wave-uniform: a tile follows
    // the tile's columns above the first row (:121: y = 0, u = q beyond the first column) and its target bases (N: 4)
    uint32_t W[kLaneTile];
    {
      const uint32_t cw = tile_on ? tw[t0 >> 4] : 0u;
      const uint32_t nm = tile_on ? tn[t0 >> 5] >> (t0 & 16) : 0u;
#pragma unroll
      for (int k = 0; k < kLaneTile; ++k) {
        const uint32_t c = ((nm >> k) & 1u) ? 4u : ((cw >> (2 * k)) & 3u);
        W[k] = ((t0 + k) ? (uint32_t)gq : 0u) | (c << 16);
      }
    }
    uint32_t qcw = 0u, qnm = 0u;
    uint32_t e_next = t0 ? edge[0] : 0u;
    for (int j = 0; j < Qw; ++j) {
      const bool row_on = j < qlen && tile_on;
      if (j < qlen) {
        if ((j & 15) == 0) qcw = qw[j >> 4];
        if ((j & 31) == 0) qnm = qn[j >> 5];
      }
      const uint32_t e_cur = e_next;
      if (t0) e_next = edge[(j + 1) * 64];  // (one row of slack behind the last)
      if (row_on) {
        const bool q_n = ((qnm >> (j & 31)) & 1u) != 0u;
        const uint32_t qc = (qcw >> ((j & 15) * 2)) & 3u;
        const int z_eq = q_n ? zwild : zm, z_ne = q_n ? zwild : zmis;
        int x = t0 ? (int)(e_cur & 0xffu) : 0, v = t0 ? (int)(e_cur >> 8) : (j ? gq : 0);  // (:120 left of the first column)
        uint32_t Fa = 0u, Fb = 0u, Fx = 0u, Fy = 0u;
        if (full) lane_row<false>(W, x, v, Fa, Fb, Fx, Fy, ncol, qc, z_eq, z_ne, zwild, cap, gq);
        else lane_row<true>(W, x, v, Fa, Fb, Fx, Fy, ncol, qc, z_eq, z_ne, zwild, cap, gq);
        if (more) edge[j * 64] = (uint16_t)((uint32_t)x | ((uint32_t)v << 8));
        if (with_dir) dirp[(size_t)(t0 >> 4) * qlen + j] = make_uint2(Fa | (Fb << 16), Fx | (Fy << 16));
        if (j == 0) {
          // exact H along the first row (:231, u read as a byte): H(0, 0) is the substitution score of the first bases
          // (:249: v - 2 (q + e) with v = z there), every further cell adds its u - (q + e)
          if (t0 == 0) hrow0 = (int)(W[0] & 0xffu) /* u(0,0) = z(0,0) */ - 2 * qe;
#pragma unroll
          for (int k = 0; k < kLaneTile; ++k)
            if (k < ncol && t0 + k > 0) hrow0 += (int)(W[k] & 0xffu) - qe;
        }
        if (last_tile) {
          // ... and down the last column (v read as a byte): rows in ascending order, the first maximum stays (:252)
          hl = j ? hl + v - qe : hrow0;
          if (hl > mte) {
            mte = hl;
            mte_j = j;
          }
          score = hl;  // (the last row's value is the score, :255)
        }
      }
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
