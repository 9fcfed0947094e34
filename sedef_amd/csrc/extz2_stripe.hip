// Register-resident extz2 DP for FULL-BAND tasks whose target is wider than one wavefront's window: the target is cut
// into stripes of NSLOT = 128 * NREG positions and every stripe is ONE wavefront (a one-wavefront workgroup) that runs
// down its anti-diagonals r' = r - T0 with its NSLOT cells in registers -- a systolic array over the stripes.
//
// A full-band task (w >= max(qlen, tlen); reference band lo = max(0, r - qlen + 1), hi = min(r, tlen - 1),
// extern/ksw2_extz2_sse.cc:101-115) has no cell whose neighbours lie outside the band except at the matrix borders:
// cell (r, t) reads x, v of (r-1, t-1) and u, y of (r-1, t), both inside the band of row r-1 unless t = 0 (the
// reference's start-of-target constants), or the cell is on the first query row (t = r: u = gap open, y = 0,
// reference :122).  So, unlike the banded kernels, nothing of the reference's 16-cell block rounding, re-basing or
// carry-in artefacts can reach a cell that is ever read: the stripe keeps its window at its first column for all rows,
// computes whole registers (what lies above the diagonal t = r or below the band start is scratch nobody reads) and
// spends its instructions on the recurrence:
//   * rows r' < tlen' ("head"): the top cell moves up one column per row along the first query row; the registers
//     up to its own are computed, the cell gets its border values before the row, and its u is added (inside the
//     owning lane) to the H of the top cell, which is handed from stripe to stripe;
//   * then all registers, every row the same instructions; the (r-1, t-1) neighbour of the stripe's first column is
//     the left stripe's last column of the previous global row, its own last column leaves for the right stripe;
//   * rows r' >= qlen ("tail"): registers wholly below the band start drop out;
//   * the last stripe owns the target's end: H of its last column is followed down the rows (inside the owning lane)
//     for mte / score.
// The stripes of a task talk through HBM (they share an XCD, so its L2): per stripe a progress word ("first query row
// done, H handed over") and, per stripe boundary, the FULL edge column: one x | v << 16 word per row of the right
// stripe that reads it (its rows 0 .. qlen - 1; row r reads what the left stripe left after its row r + NSLOT - 1).  A stripe
// stores the 16 edge words of a 16-row block with one instruction, bit 0 of a word set (the state values are
// multiples of 256: the bit is free; the columns are zeroed before the launch); its right neighbour fetches the 16
// words a block needs with one load -- issued one block ahead -- and looks at the tag bits.  A stripe only ever waits
// for its left neighbour, which has the smaller workgroup index on the same XCD: resident or finished.
// Launch-order entries are (stripe << 24) | task; stripe index 255 marks an entry that does nothing.
// Direction flags: the wave-kernel bit blocks, one region per stripe, slot = t - T0 (traceback layout 3).
//
// Compiled inside sdf_unity.hip after extz2_wave.hip (helpers, SDF_CORE, pool_code16).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "sdf_internal.h"

namespace sdf {

// bytes of LDS of one stripe's wavefront (the reversed query, byte pairs, NSLOT entries of margin either side)
constexpr int kStripeMaxT = 254 * 128;  // widest target: a launch entry has eight bits of stripe index, 255 = idle (sdf_plan.hip)
__host__ __device__ inline size_t stripe_lds_bytes(int qlen, int nreg) {
  return ((size_t)2 * (size_t)(qlen + 256 * nreg) + 15) & ~(size_t)15;
}
// bytes of a task's direction flags (one region per stripe) / of its sync words and edge columns behind them
__host__ __device__ inline size_t stripe_dir_bytes(int qlen, int tlen, int nreg) {
  const int nslot = 128 * nreg, nst = (tlen + nslot - 1) / nslot;
  return (size_t)nst * ((size_t)((qlen + nslot - 1 + 15) / 16) * nreg * 1024);
}
__host__ __device__ inline size_t stripe_sync_bytes(int qlen, int tlen, int nreg) {
  const int nslot = 128 * nreg, nst = (tlen + nslot - 1) / nslot;
  return (((size_t)nst * 8 + 255) & ~(size_t)255) + (size_t)(nst > 1 ? nst - 1 : 0) * (size_t)(qlen + 16) * 4;
}

// The words stripes exchange: relaxed atomics of agent scope -- coherent across the XCDs' L2 caches, so that the protocol
// does not depend on a task's stripes sharing an XCD (that is a matter of speed: then the words stay in its L2).
template <typename T>
__device__ __forceinline__ T ld_agent(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A wait gives up after `spin_cap` polls (context field, SDF_STRIPE_SPIN_CAP in the environment of sdf_create; 2^24 by
// default: seconds).  With claimed entries (stripe_claim below) a stripe's left neighbour has always been taken by a
// workgroup that is running or done; taken by index (SDF_STRIPE_CLAIM=0, or a launch without counters) the protocol's forward
// progress rests on the dispatch order, which a part with another XCD count, or other launches holding the wavefront slots,
// may not honour -- and a wavefront preempted for good would stall its right neighbours either way.  The
// wavefront then marks its TASK as abandoned (n_cigar = -1 in the task's result record, which nothing else touches before
// the traceback; the task's index is appended to the list behind the give-up counter, once per task) and ends; the other
// stripes of the task see the mark within 64 polls and end too.  The batch call re-runs abandoned tasks on the
// one-wavefront / one-workgroup kernels (sdf_launch.hip: rerun_abandoned).
#define SDF_GAVEUP_LIST 32   // (in 64-bit words behind the counter: the list of abandoned tasks, 32-bit entries)
#define SDF_GAVEUP_CAP 65536  // entries of the list (beyond it the batch call fails: "more tasks than can be re-run")
#define SDF_MISC_PARTS (1 + SDF_GAVEUP_LIST + SDF_GAVEUP_CAP / 2)  // (64-bit words of the context's misc buffer before the scan's partial sums)
__device__ __forceinline__ void stripe_abandon(unsigned long long *gave_up, sdf_result *rec, int out_idx, int lane) {
  if (lane != 0) return;
  const int old = __hip_atomic_exchange(&rec->n_cigar, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == -1) return;
  const unsigned long long slot = atomicAdd(gave_up, 1ull);
  if (slot < SDF_GAVEUP_CAP) reinterpret_cast<uint32_t *>(gave_up + SDF_GAVEUP_LIST)[slot] = (uint32_t)out_idx;
}
__device__ __forceinline__ bool stripe_abandoned(const sdf_result *rec) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(&rec->n_cigar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == -1;
}

// A workgroup's entry of the launch order.  The planner deals a launch's tasks to eight lists -- order[8 p + q] is entry p of
// list q -- so that the stripes of a task share an XCD's L2 (their edge words) and every stripe's left neighbour stands
// earlier in its list.  With `claim` (eight counters, zero at launch) a workgroup TAKES the next entry of the list of the
// XCD it finds itself on (HW_REG_XCC_ID), or of the next list that has one left: whatever the dispatcher does -- other
// launches' workgroups in between, another start XCD --, a task's stripes meet in one L2, and the neighbour a stripe waits
// for has been taken before it, by a workgroup that is running or done (forward progress no longer rests on the dispatch
// order).  Without `claim`: entry blockIdx.x, which is the same thing when workgroup i lands on XCD i mod 8.
// Debug (sdf_debug_placement): wavefronts started per (XCD, shader engine, CU, SIMD) by the kernels that call place_note().
// (What it showed, profiles/placement_probe.py: the 5,600 chain wavefronts of the hg19 mixture's heavy chunk land 4..7 to a SIMD
// -- mean 5.5 --, and a chain runs at the pace of its slowest block.  Workgroups of four wavefronts, one per SIMD of a CU, were
// tried against it: the dispatcher spreads them LESS evenly over the CUs, 35..56 per CU against 40..50, and the batch took
// 15.5 instead of 14.3-14.7 ms.)
__device__ unsigned *g_place = nullptr;
__device__ __forceinline__ void place_note() {
  unsigned *p = g_place;
  if (p && threadIdx.x == 0) {
    const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID: SIMD 5:4, CU 11:8, SE 15:13
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    atomicAdd(p + ((xcc << 9) | (((hw >> 13) & 7u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u)), 1u);
  }
}

__device__ __forceinline__ int32_t stripe_claim(const int32_t *__restrict__ order, unsigned *__restrict__ claim) {
  if (!claim) return order[blockIdx.x];
  int32_t entry = (int32_t)(255u << 24);  // (idle)
  if (threadIdx.x == 0) {
    const unsigned per_list = gridDim.x / 8;
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID, bits 3:0
    for (unsigned a = 0; a < 8; ++a) {
      const unsigned q = (xcc + a) & 7u;
      const unsigned p = atomicAdd(&claim[q], 1u);
      if (p < per_list) {
        entry = order[8 * p + q];
        break;
      }
    }
  }
  return __builtin_amdgcn_readfirstlane(entry);  // (one-wavefront workgroups)
}

template <int NREG>
__global__ __launch_bounds__(64, 2) void extz2_stripe_kernel(const PlanTask *__restrict__ plan,
                                                             const int32_t *__restrict__ order,
                                                             const uint32_t *__restrict__ pool, ScoreK sc,
                                                             uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res,
                                                             const int rmax, unsigned long long *__restrict__ gave_up,
                                                             const int spin_cap, unsigned *__restrict__ claim) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NSLOT = 128 * NREG;  // stripe width
  constexpr int KT = NREG - 1;
  const int32_t entry = stripe_claim(order, claim);
  const PlanTask tk = plan[entry & 0xffffff];
  const int lane = threadIdx.x;
  const int sb = (int)((uint32_t)entry >> 24);  // stripe of this wavefront
  const int tlen_all = tk.tlen;
  const int nstripe = (tlen_all + NSLOT - 1) / NSLOT;
  if (sb >= nstripe) return;  // (a padding entry of the launch order)
  const int qlen = tk.qlen;
  const int T0 = sb * NSLOT;                                       // first target position of the stripe
                                     // first target position of the stripe
  const int tlen = tlen_all - T0 < NSLOT ? tlen_all - T0 : NSLOT;  // its slice
  const int nrow = qlen + tlen - 1;                                // its anti-diagonals
  const int ncol = qlen + 16;                                      // entries of an edge column
  const bool has_left = sb > 0, has_right = sb + 1 < nstripe;
  // A launch ends with its longest task, whose stripes form ONE chain of qlen + tlen dependent rows: the wavefronts of
  // the long tasks get the SIMD first (issue priority by the task's share of the launch's longest chain), the short
  // tasks fill the gaps.
  {
    const int share = 4 * (qlen + tlen_all) / (rmax > 0 ? rmax + 1 : 1);
    if (share >= 3) __builtin_amdgcn_s_setprio(3);
    else if (share == 2) __builtin_amdgcn_s_setprio(2);
    else if (share == 1) __builtin_amdgcn_s_setprio(1);
  }
  uint8_t *gsync = dirbase + tk.dir_off + (int64_t)stripe_dir_bytes(qlen, tlen_all, NREG);
  int *prog = reinterpret_cast<int *>(gsync);  // [nstripe] T0 + NSLOT - 1 once stripe s has handed over
  int *hand_val = prog + nstripe;                       // [nstripe] H of its top cell after its row NSLOT - 1
  uint32_t *rings = reinterpret_cast<uint32_t *>(gsync + (((size_t)nstripe * 8 + 255) & ~(size_t)255));
  uint32_t *ring_out = rings + (size_t)sb * ncol;                         // [r - (NSLOT - 1)], r: my row
  uint32_t *ring_in = rings + (size_t)(has_left ? sb - 1 : 0) * ncol;  // [r], r: my row

  // ---- unpack: the reversed query as byte pairs; W[i] = (QR[i - NSLOT], QR[i - NSLOT + 1]), QR[e] = query[qlen-1-e]
  // (0 outside): lane l of register k reads entry qlen - 1 - r + NSLOT + 128 k + 2 l on row r ----
  uint16_t *W = reinterpret_cast<uint16_t *>(lds);
  bool has_n;
  unsigned TA[NREG], TB[NREG];  // score tables of the lane's two target positions per register
  const unsigned t_mis4 = (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff) * 0x01010101u, t_wild4 = (unsigned)sc.qe2_b * 0x01010101u;
  const unsigned t_delta = (unsigned)((sc.sc_match + sc.qe2_b) & 0xff) ^ (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff);
  {
    const uint32_t *tw = pool + tk.t_word, *tn = tw + (tlen_all + 15) / 16;
    const uint32_t *qw = pool + tk.q_word, *qn = qw + (qlen + 15) / 16;
    uint32_t n_seen = 0;
    for (int k = lane; k < (tlen_all + 31) / 32; k += 64) n_seen |= tn[k];
    for (int k = lane; k < (qlen + 31) / 32; k += 64) n_seen |= qn[k];
    has_n = __builtin_amdgcn_readfirstlane((int)__any(n_seen != 0)) != 0;  // wave-uniform
    const int qcap = qlen + 2 * NSLOT;
    for (int i = lane; i < qcap; i += 64) {
      const int e0 = i - NSLOT, e1 = e0 + 1;
      uint32_t v0 = (e0 >= 0 && e0 < qlen) ? pool_code16(qw, qn, qlen - 1 - e0, sc.wild) : 0u;
      uint32_t v1 = (e1 >= 0 && e1 < qlen) ? pool_code16(qw, qn, qlen - 1 - e1, sc.wild) : 0u;
      W[i] = qsel_pair(v0, v1);  // (selector form: extz2_wave.hip, SDF_SCORE2)
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int t = 128 * k + 2 * lane;
      const uint32_t c0 = t < tlen ? pool_code16(tw, tn, T0 + t, sc.wild) : 0u;
      const uint32_t c1 = t + 1 < tlen ? pool_code16(tw, tn, T0 + t + 1, sc.wild) : 0u;
      TA[k] = score_table(c0, t_mis4, t_delta, t_wild4);
      TB[k] = score_table(c1, t_mis4, t_delta, t_wild4);
    }
  }
  __syncthreads();

  // ---- constants of the <<8 difference domain ----
  const unsigned qv = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned capv = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned z_wild = ((unsigned)sc.qe2_b << 8) * 0x00010001u;  // score 0
  unsigned one2 = 0x00010001u;  // min(x, 1) per half; opaque so that it stays one v_pk_min_u16
  SDF_OPQ(one2);

  unsigned U[NREG], V[NREG], X[NREG], Y[NREG], S[NREG];
  unsigned Fa[NREG], Fb[NREG], Fx[NREG], Fy[NREG];
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
    U[k] = V[k] = X[k] = Y[k] = 0u;
    S[k] = z_wild;
    Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
  }
  const bool with_dir = !(tk.flag & SDF_FLAG_SCORE_ONLY);
  uint4 *dir = reinterpret_cast<uint4 *>(dirbase + tk.dir_off +
                                         (int64_t)sb * ((int64_t)((qlen + NSLOT - 1 + 15) / 16) * NREG * 1024));

  // H of the top cell: along the first query row it is the left stripe's (or the matrix corner's) value plus the u of
  // every border cell passed, added up inside the lanes and folded at the end of the head
  int32_t h_head = -sc.qe;  // (global row 0: H = u' - 2 (q + e), the same form with this start value)
  if (has_left) {           // the left stripe has finished global row T0 - 1: the H of its top cell is there
    int spins = 0;
    // (readfirstlane: the compiler cannot see that a volatile load of one address is wave-uniform, and a divergent
    // loop here would make every value that lives across it -- the row counters -- a vector value)
    bool lost = false;
    while (__builtin_amdgcn_readfirstlane(ld_agent(prog + sb - 1)) < T0 - 1) {
      if (++spins >= spin_cap || ((spins & 63) == 63 && stripe_abandoned(res + tk.out_idx))) {
        lost = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (lost) {  // (the batch call re-runs the task on another kernel)
      stripe_abandon(gave_up, res + tk.out_idx, tk.out_idx, lane);
      return;
    }
    // (the hand-over value was stored before the progress word, release fence in between: acquire on this side)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    h_head = __builtin_amdgcn_readfirstlane(ld_agent(hand_val + sb - 1));
  }
#ifdef SDF_STRIPE_TIMING
  unsigned long long tm_start = __builtin_amdgcn_s_memrealtime(), tm_head = 0, tm_tail = 0, tm_wait = 0;
#endif
  unsigned hacc = 0u;
  // the last stripe: H of the last column (target end), row by row from the end of the head on, inside lane LT
  const int shT = (((tlen - 1) & 1) << 4) + 8;
  int32_t ht = 0, best = SDF_NEG_INF, best_r = -1;

  // edge words: this stripe's of the current block (export side); the left neighbour's of the current block and,
  // fetched ahead, of the next one
  uint32_t edge16 = 0u, feed16 = 0u, feed_next = 0u;
  int feed_g0 = -0x40000000;
  auto feed_load = [&](const int rfirst) -> uint32_t {  // the edge words rows rfirst + (0 .. 15) read
    // a row reads the column only while its first cell is in the band (r <= qlen - 1): what lies beyond is never
    // written by the neighbour and never used here -- "tagged", value 0
    const int i = lane & 15, r = rfirst + i;
    return (r <= qlen - 1) ? ld_agent(ring_in + r) : 1u;
  };

  unsigned qaddr = 0u, qnext[NREG];
#pragma unroll
  for (int k = 0; k < NREG; ++k) qnext[k] = 0u;

  // ------------------------------------------------------------------------------------------
  // Rows [rb, re) of one 16-row block with the registers KLO .. KHI active.
  //   head: these are rows of the first query row's passage (r <= tlen - 1): border values for the cell t = r
  //         (register KHI) before the row, its u into the H sum after it;
  //   exp:  the last slot's x, v leave as this stripe's edge word (rows >= NSLOT - 1 of a stripe with a right
  //         neighbour; KHI == KT then);
  //   trk:  the last stripe after its head: H of the last column (register KHI).
  // The flags are constant over the call (the caller cuts the block at r = tlen and r = NSLOT - 1).
  // ------------------------------------------------------------------------------------------
  // (FLAGS: the three flags as constants -- head | exp << 1 | trk << 2 -- for the narrower stripe widths, whose rows
  // are short enough for the flag branches to show; -1: read at run time, as the 512-position width does for all but its
  // all-register rows, to keep its ten register ranges from growing into sixty row loops)
  auto rows = [&](auto klo_c, auto khi_c, auto flags_c, const bool head_rt, const bool exp_rt, const bool trk_rt, const int rb,
                  const int re) {
    constexpr int KLO = decltype(klo_c)::value, KHI = decltype(khi_c)::value, FLAGS = decltype(flags_c)::value;
    const bool head = FLAGS < 0 ? head_rt : (FLAGS & 1) != 0;
    const bool exp = FLAGS < 0 ? exp_rt : (FLAGS & 2) != 0;
    const bool trk = FLAGS < 0 ? trk_rt : (FLAGS & 4) != 0;
    qaddr = (unsigned)(2 * (qlen - 1 - rb + NSLOT + 2 * lane));
#pragma unroll
    for (int k = KLO; k <= KHI; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
    const bool feed = KLO == 0 && has_left;
    int fidx = rb + T0 - 1 - feed_g0;  // lane of feed16 with the left edge word of the previous global row
    int eidx = rb & 15;                // lane of edge16 this row's edge word goes to
#pragma unroll 1
    for (int r = rb; r < re; ++r, ++fidx, ++eidx) {
      // query codes of this row (fetched during the previous one), bytes -> halves; the next row's fetch is issued
      // only now -- behind the use of the last one, so that the wait for LDS above finds it long complete instead of
      // covering a fetch issued just before it (the empty asm ties the address to the consumed values)
      unsigned qc[NREG];
#pragma unroll
      for (int k = KLO; k <= KHI; ++k) qc[k] = qsel_spread(qnext[k]);
      qaddr -= 2;
      asm volatile("" : "+v"(qaddr) : "v"(qc[KLO]), "v"(qc[KHI]));
#pragma unroll
      for (int k = KLO; k <= KHI; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
      const int sr = r - 128 * KHI;  // slot of the cell t = r inside register KHI (head rows)
      const bool mine = lane == (sr >> 1);
      if (head) {  // border cell t = r: y = 0, u = gap open (reference :122; 0 on global row 0)
        const unsigned keep = (sr & 1) ? 0x0000ffffu : 0xffff0000u;
        const unsigned uval = (r + T0) ? (((unsigned)sc.q_b << 8) << ((sr & 1) * 16)) : 0u;
        U[KHI] = mine ? ((U[KHI] & keep) | uval) : U[KHI];
        Y[KHI] = mine ? (Y[KHI] & keep) : Y[KHI];
      }
      // (r-1, t-1) neighbours: x and v one slot up; into slot 0 the left stripe's edge word of the previous global
      // row, or the start-of-target constants (x = 0, v = gap open; 0 on row 0)
      unsigned xt1[NREG], vt1[NREG];
#pragma unroll
      for (int k = KLO; k <= KHI; ++k) {
        unsigned xs, vs;
        if (k == 0) {
          unsigned xc = 0u, vc = r ? ((unsigned)sc.q_b << 24) : 0u;
          if (feed) {
            const uint32_t fe = (uint32_t)__builtin_amdgcn_readlane((int)feed16, fidx);
            xc = (fe & 0xfffeu) << 16;
            vc = fe & 0xffff0000u;
          }
          xs = (unsigned)__builtin_amdgcn_update_dpp((int)xc, (int)X[0], 0x138, 0xf, 0xf, false);
          vs = (unsigned)__builtin_amdgcn_update_dpp((int)vc, (int)V[0], 0x138, 0xf, 0xf, false);
        } else {
          // (register k - 1 may have dropped out of the band: its last values are those of the row before it did,
          // which is the only row on which slot 128 k still reads them)
          int ux, uv;  // lanes 1..63 are overwritten by the second move: no initial value needed
          asm("" : "=v"(ux));
          asm("" : "=v"(uv));
          const int x0 = __builtin_amdgcn_update_dpp(ux, (int)X[k - 1], 0x13C, 0x1, 0x1, false);
          xs = (unsigned)__builtin_amdgcn_update_dpp(x0, (int)X[k], 0x138, 0xf, 0xf, false);
          const int v0 = __builtin_amdgcn_update_dpp(uv, (int)V[k - 1], 0x13C, 0x1, 0x1, false);
          vs = (unsigned)__builtin_amdgcn_update_dpp(v0, (int)V[k], 0x138, 0xf, 0xf, false);
        }
        xt1[k] = __builtin_amdgcn_alignbit(X[k], xs, 16);
        vt1[k] = __builtin_amdgcn_alignbit(V[k], vs, 16);
      }
      // scores of the row's cells
#pragma unroll
      for (int k = KLO; k <= KHI; ++k) SDF_SCORE2(S[k], k, qc[k], false)
      if (has_n) {  // (an N in the query: the selector picked 0xff)
#pragma unroll
        for (int k = KLO; k <= KHI; ++k) {
          unsigned nn = pk_ashr15(S[k]);
          SDF_OPQ(nn);
          S[k] = (z_wild & nn) | (S[k] & ~nn);
        }
      }
#pragma unroll
      for (int k = KLO; k <= KHI; ++k) SDF_CORE(k)
      if (head) hacc += mine ? ((U[KHI] >> (((sr & 1) << 4) + 8)) & 0xffu) : 0u;
      if (exp) {
        // x | v << 16 of the last slot (high halves of lane 63), tagged, into its lane of the block's edge words
        const unsigned ew = __builtin_amdgcn_perm(V[KT], X[KT], 0x07060302u);
        const unsigned es = (unsigned)__builtin_amdgcn_readlane((int)ew, 63) | 1u;  // (outside the select: all lanes)
        edge16 = lane == eidx ? es : edge16;
      }
      if (trk) {  // H(r, T) = H(r-1, T) + v(r, T), T = tlen - 1
        ht += (int32_t)((V[KHI] >> shT) & 0xffu) - sc.qe;
        const bool gt = ht > best;
        best = gt ? ht : best;
        best_r = gt ? r : best_r;
      }
    }
  };
  auto run_rows = [&](const int klo, const int khi, const bool head, const bool exp, const bool trk, const int rb,
                      const int re) {
#define SDF_ROWS_F(A, B, F) \
  rows(std::integral_constant<int, (A)>{}, std::integral_constant<int, (B)>{}, std::integral_constant<int, (F)>{}, head, exp, trk, rb, re)
#define SDF_ROWS(A, B)                                                                                               \
  case (A) * 4 + (B):                                                                                                 \
    if constexpr ((B) < NREG && (A) <= (B)) {                                                                         \
      if constexpr (NREG >= 4 && !((A) == 0 && (B) == NREG - 1)) { /* (512 positions: the all-register rows only) */ \
        SDF_ROWS_F(A, B, -1);                                                                                         \
      } else {                                                                                                        \
        switch ((head ? 1 : 0) | (exp ? 2 : 0) | (trk ? 4 : 0)) { /* (exp and trk exclude each other) */             \
          case 0: SDF_ROWS_F(A, B, 0); break;                                                                         \
          case 1: SDF_ROWS_F(A, B, 1); break;                                                                         \
          case 2: SDF_ROWS_F(A, B, 2); break;                                                                         \
          case 3: SDF_ROWS_F(A, B, 3); break;                                                                         \
          case 4: SDF_ROWS_F(A, B, 4); break;                                                                         \
          default: SDF_ROWS_F(A, B, -1); break;                                                                       \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
    break;
    switch (klo * 4 + khi) {
      SDF_ROWS(0, 0) SDF_ROWS(0, 1) SDF_ROWS(0, 2) SDF_ROWS(0, 3) SDF_ROWS(1, 1) SDF_ROWS(1, 2) SDF_ROWS(1, 3)
      SDF_ROWS(2, 2) SDF_ROWS(2, 3) SDF_ROWS(3, 3)
      default: break;
    }
#undef SDF_ROWS
#undef SDF_ROWS_F
  };

  for (int r0 = 0; r0 < nrow; r0 += 16) {
    r0 = __builtin_amdgcn_readfirstlane(r0);  // (wave-uniform; stated, so that the row loops stay scalar branches)
    const int rend = r0 + 16 < nrow ? r0 + 16 : nrow;
    // registers with a cell of the band on some row of the block
    const int lo_first = r0 - qlen + 1 > 0 ? r0 - qlen + 1 : 0;
    const int hi_last = rend - 1 < tlen - 1 ? rend - 1 : tlen - 1;
    const int klo = lo_first >> 7, khi = hi_last >> 7;
    if (has_left && klo == 0) {
      // the sixteen edge words of the block: fetched during the previous block; all tagged, or fetched again
      uint32_t got = (feed_g0 + 16 == T0 + r0 - 1) ? feed_next : feed_load(r0);
      int spins = 0;
#ifdef SDF_STRIPE_TIMING
      const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
      while (__builtin_amdgcn_readfirstlane((int)__any((got & 1u) == 0u))) {
        if (++spins >= spin_cap || ((spins & 63) == 63 && stripe_abandoned(res + tk.out_idx))) {
          stripe_abandon(gave_up, res + tk.out_idx, tk.out_idx, lane);
          return;
        }
        __builtin_amdgcn_s_sleep(2);
        got = feed_load(r0);
      }
#ifdef SDF_STRIPE_TIMING
      if (spins) tm_wait += __builtin_amdgcn_s_memrealtime() - tw0;
#endif
      feed16 = got;
      feed_g0 = T0 + r0 - 1;
      feed_next = feed_load(r0 + 16);  // for the next block; checked there
    }
    // cut at the end of the head (r = tlen) and where the edge export starts (r = NSLOT - 1)
    int r = r0;
    while (r < rend) {
      r = __builtin_amdgcn_readfirstlane(r);
      const bool head = r < tlen;
      const bool exp = has_right && r >= NSLOT - 1;
      int stop = rend;
      if (head && tlen < stop) stop = tlen;
      if (has_right && r < NSLOT - 1 && NSLOT - 1 < stop) stop = NSLOT - 1;
      run_rows(klo, khi, head, exp, !has_right && !head, r, stop);
      r = stop;
      if (r == tlen) {  // the first query row has reached the end of the slice: fold the H sum
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) hacc += (unsigned)__shfl_xor((int)hacc, off);
        h_head += (int32_t)hacc - tlen * sc.qe;
        ht = best = h_head;  // (the last stripe: H(tlen - 1, tlen - 1), the first candidate for mte)
        best_r = tlen - 1;
#ifdef SDF_STRIPE_TIMING
        tm_head = __builtin_amdgcn_s_memrealtime();
#endif
      }
#ifdef SDF_STRIPE_TIMING
      if (!tm_tail && r >= qlen) tm_tail = __builtin_amdgcn_s_memrealtime();
#endif
    }
    // ---- block end: direction flags of these (<= 16) rows leave for HBM ----
    if (with_dir) {
      const int done = rend - r0;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (k < klo || k > khi) continue;
        unsigned fa = Fa[k], fb = Fb[k], fx = Fx[k], fy = Fy[k];
        if (done < 16) {
          const unsigned sh = 16 - done;
          fa = pk_shl(fa, sh);
          fb = pk_shl(fb, sh);
          fx = pk_shl(fx, sh);
          fy = pk_shl(fy, sh);
        }
        dir[((int64_t)(r0 >> 4) * NREG + k) * 64 + lane] = make_uint4(fa, fb, fx, fy);
      }
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    if (has_right) {
      if (rend > NSLOT - 1) {  // this block's edge words (the tagged ones: rows >= NSLOT - 1)
        const int g = r0 + lane - (NSLOT - 1);
        if (lane < 16 && (edge16 & 1u) && g < ncol) st_agent(ring_out + g, edge16);
      }
      edge16 = 0u;
      if (rend == NSLOT) {  // first query row done, its edge word stored: the right neighbour may start
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) {
          st_agent(hand_val + sb, h_head);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          st_agent(prog + sb, T0 + NSLOT - 1);
        }
      }
    }
  }
#ifdef SDF_STRIPE_TIMING
  if (lane == 0)
    printf("stripe %d start %llu head %llu tail %llu end %llu wait %llu\n", sb, tm_start, tm_head, tm_tail,
           (unsigned long long)__builtin_amdgcn_s_memrealtime(), tm_wait);
#endif
  if (has_right) return;  // the last stripe owns the end of the target: score, mte

  const int lt = ((tlen - 1) & 127) >> 1;
  const int32_t score = __builtin_amdgcn_readlane(ht, lt);
  const int32_t mte = __builtin_amdgcn_readlane(best, lt);
  const int32_t mte_r = __builtin_amdgcn_readlane(best_r, lt);
  if (lane == 0) {
    sdf_result o;
    o.score = score;
    o.max = 0;
    o.max_q = o.max_t = -1;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = mte;
    o.mte_q = mte_r - ((tlen - 1) | 15);  // (the reference's en is block-rounded, :209)
    o.zdropped = 0;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[tk.out_idx] = o;
  }
}

template __global__ void extz2_stripe_kernel<1>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                sdf_result *, int, unsigned long long *, int, unsigned *);
template __global__ void extz2_stripe_kernel<2>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                sdf_result *, int, unsigned long long *, int, unsigned *);
template __global__ void extz2_stripe_kernel<4>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                sdf_result *, int, unsigned long long *, int, unsigned *);

// Before the launch, one workgroup per launch-order entry (task, stripe): the stripe's progress and hand-over words to
// "nothing done" and the edge column of its right boundary to zero (no word tagged as written)
__global__ __launch_bounds__(64) void stripe_sync_init_kernel(const PlanTask *__restrict__ plan,
                                                              const int32_t *__restrict__ order, int nreg,
                                                              uint8_t *__restrict__ dirbase) {
  const int32_t entry = order[blockIdx.x];
  const PlanTask tk = plan[entry & 0xffffff];
  const int sb = (int)((uint32_t)entry >> 24);
  const int nslot = 128 * nreg, nst = (tk.tlen + nslot - 1) / nslot;
  if (sb >= nst) return;
  uint8_t *gsync = dirbase + tk.dir_off + (int64_t)stripe_dir_bytes(tk.qlen, tk.tlen, nreg);
  int *prog = reinterpret_cast<int *>(gsync);
  if (threadIdx.x == 0) {
    prog[sb] = -1;
    prog[nst + sb] = 0;
  }
  if (sb + 1 < nst) {
    const int ncol = tk.qlen + 16;
    uint32_t *col = reinterpret_cast<uint32_t *>(gsync + (((size_t)nst * 8 + 255) & ~(size_t)255)) + (size_t)sb * ncol;
    for (int g = threadIdx.x; g < ncol; g += 64) col[g] = 0u;
  }
}

}  // namespace sdf
