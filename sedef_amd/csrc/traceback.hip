// Traceback (direction matrix -> CIGAR + column counts) and CIGAR compaction for gfx950.
//
// Restates ksw_backtrack / ksw_push_cigar (reference: extern/ksw2.h:98-151, rotated layout)
// and the counters of populate_nice_alignment (reference: src/align.cc:274-315).
//
// The reference's walk is serial per task; here a group of lanes walks one task a whole RUN at a time (see
// traceback_kernel below).
#include <hip/hip_runtime.h>

#include "sdf_internal.h"

namespace sdf {

__device__ __forceinline__ uint32_t code_at(const uint32_t *codes, const uint32_t *nmask, int k) {
  const uint32_t c = (codes[k >> 4] >> ((k & 15) * 2)) & 3u;
  const uint32_t n = (nmask[k >> 5] >> (k & 31)) & 1u;
  return n ? 4u : c;
}

// ---- one cell of the direction matrix ----------------------------------------------------------------------
// LAYOUT: 0 = byte rows (general kernel), 1 = wave-kernel bit blocks, 2 = pair-kernel bit blocks, 3 = wave-kernel
// bit blocks per target stripe (extz2_stripe.hip).  A cell outside the block range the reference stored for its
// anti-diagonal has no flags: the walk is forced there (reference: extern/ksw2.h:128-130) -- state 2 below the
// range, 1 above it.
struct TbAddr {
  int64_t idx;    // record to load (0 when nothing is to be loaded)
  uint32_t meta;  // bit position of the row inside the record | 0x100 | forced state
};

template <int LAYOUT>
__device__ __forceinline__ TbAddr tb_addr(const PlanTask &tk, int i, int j) {
  const int r = i + j;
  Band b;
  band_of(r, tk.qlen, tk.tlen, tk.w, b);
  TbAddr a;
  if (i < b.lo || i > b.hi) {
    a.idx = 0;
    a.meta = 0x100u | (i < b.lo ? 2u : 1u);
    return a;
  }
  if (LAYOUT == 0) {
    a.idx = (int64_t)r * tk.ncol16 + (i - b.lo);
    a.meta = 0;
  } else if (LAYOUT == 6) {
    // strip kernel (extz2_strip.hip): per step and lane one word of this task (the records are 8 bytes: its partner's
    // word sits beside it), step = the row + the lane, a block of 512 columns after the other
    // (nreg: the columns per lane, 8 or 4; ncol16: the rows of the wavefront's larger task, bit 30: the task has the
    // wavefront to itself -- records of one word)
    const int sh = tk.nreg == 8 ? 3 : 2, il = i & ((64 << sh) - 1), ln = il >> sh;
    // (four columns per lane: one record per pair of steps, the odd step's columns in the high bits of every byte)
    const int rows = tk.ncol16 & 0x3fffffff, st = j + ln;
    const int nrec = tk.nreg == 8 ? rows + 63 : (rows + 64) >> 1, rec = tk.nreg == 8 ? st : st >> 1;
    a.idx = (((int64_t)(i >> (6 + sh)) * nrec + rec) * 64 + ln) * ((tk.ncol16 >> 30) ? 1 : 2);
    a.meta = (uint32_t)(tk.nreg - 1 - (il & (tk.nreg - 1))) + (tk.nreg == 8 ? 0u : (uint32_t)(st & 1) << 2);
  } else if (LAYOUT == 5) {
    // lane kernel (extz2_lane.hip): per tile of 16 target positions one record per query position -- the pair kernels'
    // (a | b << 16, x | y << 16) with the tile's column k at bit 15 - k of every half
    a.idx = (int64_t)(i >> 4) * tk.qlen + j;
    a.meta = (uint32_t)(15 - (i & 15));
  } else if (LAYOUT == 4) {
    // banded stripes (extz2_bstripe.hip): a flag region per stripe of 128 * nreg target positions, 16-row blocks counted
    // from the block of the stripe's first row, slot t - T0
    const BStripeGeom g = bstripe_geom(tk.qlen, tk.tlen, tk.w, tk.nreg);
    const int sb = i / g.nslot, t0 = sb * g.nslot;
    const int rb = (r >> 4) - (bstripe_first_row(t0, tk.w) >> 4);
    const int slot = i - t0;
    a.idx = (int64_t)sb * (int64_t)(g.flag_bytes / 16) + (int64_t)rb * (tk.nreg * 64) + (slot >> 1);
    a.meta = (uint32_t)(15 - (r & 15) + ((slot & 1) << 4));
  } else if (LAYOUT == 3) {
    // stripes of 128 * nreg target positions, each with its own flag region in local coordinates (row r - T0,
    // slot t - T0: a stripe's window stays at its first column, extz2_stripe.hip)
    const int sw = 128 * tk.nreg;
    const int sb = i / sw, t0 = sb * sw;
    const int rp = r - t0, rb = rp >> 4;
    const int slot = i - t0;
    const int per_stripe = ((tk.qlen + sw - 1 + 15) / 16) * tk.nreg * 64;  // uint4 records
    a.idx = (int64_t)sb * per_stripe + rb * (tk.nreg * 64) + (slot >> 1);  // (register k, lane l) = slots 128k + 2l, +1
    a.meta = (uint32_t)(15 - (rp & 15) + ((slot & 1) << 4));
  } else {
    const int rb = r >> 4;
    Band b0;
    band_of(rb << 4, tk.qlen, tk.tlen, tk.w, b0);  // window base of the 16-row block
    const int slot = i - b0.lo;
    if (LAYOUT == 2) {  // per register k (64 slots) and lane one uint2 (a | b << 16, x | y << 16) of 16-bit row masks
      a.idx = (int64_t)rb * (tk.nreg * 64) + slot;
      a.meta = (uint32_t)(15 - (r & 15));
    } else {  // per register k and lane one uint4 of 32-bit words (a>z, b>z', x>0, y>0); +16 for the odd slot
      a.idx = (int64_t)rb * (tk.nreg * 64) + (slot >> 1);
      a.meta = (uint32_t)(15 - (r & 15) + ((slot & 1) << 4));
    }
  }
  return a;
}

// the reference's direction byte of the cell (bits 0-1: 0 diagonal / 1 E / 2 F won, bit 3: E continues, bit 4: F
// continues), or 0x100 | forced state
template <int LAYOUT>
__device__ __forceinline__ uint32_t tb_load(const uint8_t *dir, const TbAddr a) {
  if (LAYOUT == 0) {
    const uint32_t d = dir[a.idx];
    return (a.meta & 0x100u) ? a.meta : d;
  } else if (LAYOUT == 6) {
    const uint32_t w = reinterpret_cast<const uint32_t *>(dir)[a.idx] >> (a.meta & 7u);
    const uint32_t fa = w & 1u, fb = (w >> 8) & 1u, fx = (w >> 16) & 1u, fy = (w >> 24) & 1u;
    return (a.meta & 0x100u) ? a.meta : ((fb ? 2u : fa) | (fx << 3) | (fy << 4));
  } else if (LAYOUT == 2 || LAYOUT == 5) {
    const uint2 c = reinterpret_cast<const uint2 *>(dir)[a.idx];
    const uint32_t bit = a.meta & 31u;
    const uint32_t fa = (c.x >> bit) & 1u, fb = (c.x >> (bit + 16)) & 1u;
    const uint32_t fx = (c.y >> bit) & 1u, fy = (c.y >> (bit + 16)) & 1u;
    return (a.meta & 0x100u) ? a.meta : ((fb ? 2u : fa) | (fx << 3) | (fy << 4));
  } else {
    const uint4 c = reinterpret_cast<const uint4 *>(dir)[a.idx];
    const uint32_t bit = a.meta & 31u;
    const uint32_t fa = (c.x >> bit) & 1u, fb = (c.y >> bit) & 1u;
    const uint32_t fx = (c.z >> bit) & 1u, fy = (c.w >> bit) & 1u;
    return (a.meta & 0x100u) ? a.meta : ((fb ? 2u : fa) | (fx << 3) | (fy << 4));
  }
}

// state of the walk at a cell entered in state `s_in` (ksw_backtrack's state machine, extern/ksw2.h:131-137)
__device__ __forceinline__ int tb_state(int s_in, uint32_t cell) {
  if (cell & 0x100u) return (int)(cell & 7u);
  return (s_in != 0 && ((cell >> (s_in + 2)) & 1u)) ? s_in : (int)(cell & 7u);
}

// ---- the walk ----------------------------------------------------------------------------------------------------
// ksw_backtrack is serial: the cell a step reads depends on the state the previous cell left.  But a walk is a
// sequence of RUNS -- diagonal steps while the cells say "diagonal", gap steps while the continuation bit is set --
// and whether a run goes on through its k-th cell depends on that cell alone.  A group of G lanes walks one task:
// the flags of the current cell are in hand (they give the state s of the run that starts here), lane k reads the
// (k+1)-th cell ahead in THAT direction -- one address, one load per lane and round --, one ballot tells how far the
// run goes (up to G cells), the match / mismatch columns of a diagonal run are counted with a second ballot, the whole
// run is emitted as one CIGAR push, and the flags of the cell the run stopped at -- some lane has just read them --
// become the current ones.  A 1000 x 1000 task at 10 % divergence is ~60 such rounds (one memory latency each) instead
// of 2,000
// dependent steps.  G = 64: one task per wavefront (few or long tasks); G = 16: four tasks per wavefront (bulk).
// Runs are emitted from the END of the task's staging slot towards its start, which leaves them in forward order.
template <int LAYOUT, int G>
__global__ __launch_bounds__(64) void traceback_kernel(const PlanTask *__restrict__ plan, int n,
                                                       const uint32_t *__restrict__ pool,
                                                       const uint8_t *__restrict__ dirbase,
                                                       sdf_result *__restrict__ res,
                                                       uint32_t *__restrict__ stage) {
  constexpr int PER = 64 / G;
  const int lane = threadIdx.x, kk = lane % G, gbase = lane - kk;
  const int k = (int)blockIdx.x * PER + lane / G;
  bool active = k < n;
  PlanTask tk = plan[active ? k : 0];
  active = active && (tk.nreg == 0 ? 0 : tk.pad_ == 2 ? 2 : tk.pad_ == 5 ? 3 : tk.pad_ == 7 ? 4 : tk.pad_ == 8 ? 5 : (tk.pad_ == 9 || tk.pad_ == 10) ? 6 : 1) == LAYOUT &&
           !(tk.flag & SDF_FLAG_SCORE_ONLY);
  sdf_result rr = res[tk.out_idx];
  active = active && rr.n_cigar != -1;  // (-1: a stripe kernel gave the task up -- nothing to walk; it is run again)
  int i = -1, j = -1;  // (sequences are shorter than 2^31)
  if (active) {
    if (!rr.zdropped && !(tk.flag & SDF_FLAG_EXTZ_ONLY)) {
      i = tk.tlen - 1;
      j = tk.qlen - 1;
    } else if (rr.max_t >= 0 && rr.max_q >= 0) {
      i = rr.max_t;
      j = rr.max_q;
    } else {
      active = false;  // n_cigar stays 0
    }
  }
  const uint8_t *dir = dirbase + tk.dir_off;
  const uint32_t *tw = pool + tk.t_word, *tn = tw + (tk.tlen + 15) / 16;
  const uint32_t *qw = pool + tk.q_word, *qn = qw + (tk.qlen + 15) / 16;
  uint32_t *slot = stage + tk.cig_slot;
  int pos = tk.cig_cap;
  int cur_op = -1, cur_len = 0;
  int32_t matches = 0, mismatches = 0, gaps = 0, gap_bases = 0;
  const uint64_t gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);

  auto flush = [&]() {
    if (cur_op >= 0) {
      --pos;
      if (kk == 0) slot[pos] = ((uint32_t)cur_len << 4) | (uint32_t)cur_op;
      if (cur_op != 0) {  // the counters of populate_nice_alignment (src/align.cc:274-315)
        ++gaps;
        gap_bases += cur_len;
      }
    }
  };
  auto push = [&](int op, int len) {
    if (op == cur_op) {
      cur_len += len;
    } else {
      flush();
      cur_op = op;
      cur_len = len;
    }
  };

  // The flags of the CURRENT cell are always in hand: loaded before the first round, afterwards they are those of the
  // cell the last run stopped at, which one of the lanes had read.  So a round knows the state s of the run that starts
  // here before it loads anything, and every lane reads ONE cell: the (k+1)-th ahead in that run's direction.
  int s_in = 0;
  uint32_t c0 = 0;
  {
    const bool live0 = active && (i | j) >= 0;
    c0 = tb_load<LAYOUT>(dir, tb_addr<LAYOUT>(tk, live0 ? i : 0, live0 ? j : 0));
  }
  while (__any(active && (i | j) >= 0)) {
    const bool live = active && (i | j) >= 0;
    const int ci = live ? i : 0, cj = live ? j : 0;
    const int s = tb_state(s_in, c0);
    // the cell kk + 1 steps ahead
    const int di = ci - (s != 2 ? kk + 1 : 0), dj = cj - (s != 1 ? kk + 1 : 0);
    const bool valid = di >= 0 && dj >= 0;
    const uint32_t cs = tb_load<LAYOUT>(dir, tb_addr<LAYOUT>(tk, valid ? di : ci, valid ? dj : cj));
    // bases of the run's own cells (kk steps ahead), diagonal runs only: match <=> neither is N and the codes agree
    // (src/align.cc:29-35 on the codes)
    const bool vb = s == 0 && ci - kk >= 0 && cj - kk >= 0;
    const uint32_t tb_ = vb ? code_at(tw, tn, ci - kk) : 4u, qb_ = vb ? code_at(qw, qn, cj - kk) : 5u;
    const bool stay = live && valid && tb_state(s, cs) == s;
    const uint64_t bits = (__ballot(stay) >> gbase) & gmask;
    const uint64_t stop = ~bits & gmask;
    // cells of the run: the current one and the leading ones that go on with it -- at most G, so that the cell the run
    // stops at (or pauses at: then the next round continues it) is one a lane has read
    const int ahead = stop ? __builtin_ctzll(stop) : G - 1;
    const int nrun = ahead + 1;
    const uint64_t mb = (__ballot(tb_ == qb_ && tb_ < 4u) >> gbase) & (nrun >= 64 ? ~0ull : ((1ull << nrun) - 1ull));
    // the flags of the cell after the run: read by lane `ahead` of the group (if it lies outside the matrix the walk
    // ends and they are not used)
    const uint32_t c_next = (uint32_t)__shfl((int)cs, gbase + ahead);
    if (live) {
      if (s == 0) {
        const int m = __popcll(mb);
        matches += m;
        mismatches += nrun - m;
        i -= nrun;
        j -= nrun;
        push(0, nrun);
      } else if (s == 1) {  // E: consumes the target (ksw op 2, D)
        i -= nrun;
        push(2, nrun);
      } else {  // F: consumes the query (ksw op 1, I)
        j -= nrun;
        push(1, nrun);
      }
      s_in = s;
      c0 = c_next;
    }
  }
  if (active) {
    if (i >= 0) push(2, i + 1);
    if (j >= 0) push(1, j + 1);
    flush();
    if (kk == 0) {
      rr.n_cigar = tk.cig_cap - pos;
      rr.matches = matches;
      rr.mismatches = mismatches;
      rr.gaps = gaps;
      rr.gap_bases = gap_bases;
      res[tk.out_idx] = rr;
    }
  }
}

#define SDF_TB_INST(L)                                                                                              \
  template __global__ void traceback_kernel<L, 16>(const PlanTask *, int, const uint32_t *, const uint8_t *,      \
                                                    sdf_result *, uint32_t *);                                     \
  template __global__ void traceback_kernel<L, 64>(const PlanTask *, int, const uint32_t *, const uint8_t *,      \
                                                    sdf_result *, uint32_t *);
SDF_TB_INST(0)
SDF_TB_INST(1)
SDF_TB_INST(2)
SDF_TB_INST(3)
SDF_TB_INST(4)
SDF_TB_INST(5)
SDF_TB_INST(6)
#undef SDF_TB_INST

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
