// Register-resident extz2 DP for LONG BANDED tasks: the target is cut into stripes of NSLOT = 128 * NREG positions and
// every stripe is one wavefront (a one-wavefront workgroup, as in extz2_stripe.hip) that owns the reference's state
// arrays u, v, x, y, s, H (extern/ksw2_extz2_sse.cc:83-85: indexed by TARGET position) for its columns, in registers.
//
// A banded task is a chain of qlen + tlen dependent anti-diagonals; a wavefront alone on its SIMD issues an
// instruction every ~5 cycles, so the time of a row is the number of instructions it takes.  The one-task kernels hold
// the whole band window (up to 1024 cells) in one wavefront or walk it with a 1024-thread workgroup through LDS:
// 1 - 2 us per row, 40 - 80 ms for a 20 kb task.  Here the band (2 w + 1 cells) lies over (2 w + 1) / NSLOT + 1
// stripes at a time, each wavefront computes its 128 * NREG columns of the row -- about a hundred instructions at
// NREG = 1 -- and the stripes follow each other one 16-row block apart.
//
// Because a stripe's lanes ARE the reference's array slots, its artefacts need no emulation of a moving window: a row
// computes the cells of the 16-cell blocks [lo, hi] the reference computes (:115), from whatever the slots hold;
// the scores are refreshed in 16-cell strides from the band start (:124-138); the (r-1, t-1) neighbour of a cell is
// the slot to its left as the previous row left it, except for the first computed cell (:140-146: the start-of-target
// constants, 0 when the window did not move, the slot to its left with the sign-extension smear when it did), and the
// slot left of the stripe's first one is the left stripe's last, handed over through HBM with the H of that column:
// one (x | v << 16 | tag, H) pair per row, stored sixteen at a time per 16-row block and fetched one block ahead, as
// in extz2_stripe.hip.  H of every band cell is kept (:222-258), with the best cell per column; a finishing kernel
// merges the stripes' bests in the reference's order and writes the result record.
// Rows of a stripe: from sixteen columns before hi(r) >= T0 (the score refresh reaches that far ahead) to the last row
// with lo(r) < T1, i.e. r in [max(T0 - 16, 2 (T0 - 16) - w), min(T1 + qlen, 2 T1 + w) - 2] (cut at the end of the matrix
// or where the band runs out).  Direction flags: bit blocks per stripe, block index relative to
// the stripe's first row block, slot = t - T0 (traceback layout 4).
//
// Compiled inside sdf_unity.hip after extz2_wave.hip and extz2_general.hip (helpers, BestCell).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "sdf_internal.h"

namespace sdf {

#ifndef SDF_BS_FIRST_SLEEP
#define SDF_BS_FIRST_SLEEP 100
#endif

struct BStripeGeom {
  int nslot, nst, blocks_cap, col_len;
  size_t flag_bytes;  // per stripe
};
__host__ __device__ inline BStripeGeom bstripe_geom(int qlen, int tlen, int w, int nreg) {
  BStripeGeom g;
  g.nslot = 128 * nreg;
  const int t16 = (tlen + 15) / 16 * 16;
  g.nst = (t16 + g.nslot - 1) / g.nslot;
  const int rows = 2 * g.nslot + 2 * w < g.nslot + qlen ? 2 * g.nslot + 2 * w : g.nslot + qlen;
  g.blocks_cap = (rows + 32 + 15) / 16 + 2;
  g.col_len = g.blocks_cap * 16 + 64;
  g.flag_bytes = (size_t)g.blocks_cap * nreg * 1024;
  return g;
}
// First / last anti-diagonal of the stripe [T0, T1) (band not cut by its end).  The last one is the last on which it
// has a computed cell; the first one is sixteen columns early: the score refresh runs in 16-cell strides from the band
// START (:124-138), so it reaches up to fifteen cells past the last computed block -- into the first columns of a
// stripe that computes nothing yet, and a cell computed later as part of a widened block may still hold that score.
__host__ __device__ inline int bstripe_first_row(int T0, int w) {
  const int t = T0 >= 16 ? T0 - 16 : 0;
  return t > 2 * t - w ? t : 2 * t - w;
}
__host__ __device__ inline int bstripe_last_row(int T1, int qlen, int tlen, int w) {
  int z = qlen + tlen - 2;
  if (z > T1 + qlen - 2) z = T1 + qlen - 2;
  if (z > 2 * T1 + w - 2) z = 2 * T1 + w - 2;
  return z;
}
// bytes of a task's direction flags / of what lies behind them: a 64-byte record per stripe, then the edge columns
__host__ __device__ inline size_t bstripe_dir_bytes(int qlen, int tlen, int w, int nreg) {
  const BStripeGeom g = bstripe_geom(qlen, tlen, w, nreg);
  return (size_t)g.nst * g.flag_bytes;
}
__host__ __device__ inline size_t bstripe_sync_bytes(int qlen, int tlen, int w, int nreg) {
  const BStripeGeom g = bstripe_geom(qlen, tlen, w, nreg);
  return (size_t)g.nst * 64 + (size_t)(g.nst > 1 ? g.nst - 1 : 0) * (size_t)g.col_len * 8;
}
__host__ __device__ inline size_t bstripe_lds_bytes(int w, int nreg) {
  return ((size_t)2 * (size_t)(6 * 128 * nreg + 2 * w + 256) + 15) & ~(size_t)15;  // reversed-query window, byte pairs
}

struct BStripeRec {  // what a stripe leaves for the finishing kernel
  int32_t bestH, bestR, bestKey, bestT;
  int32_t score, mte, mte_q, flags;  // flags: 1 score set, 2 record written
  int32_t pad[8];
};

template <int NREG>
__global__ __launch_bounds__(64, 2) void extz2_bstripe_kernel(const PlanTask *__restrict__ plan,
                                                              const int32_t *__restrict__ order,
                                                              const uint32_t *__restrict__ pool, ScoreK sc,
                                                              uint8_t *__restrict__ dirbase,
                                                              sdf_result *__restrict__ res,
                                                              unsigned long long *__restrict__ gave_up,
                                                              const int spin_cap, unsigned *__restrict__ claim) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NSLOT = 128 * NREG;
  constexpr int KT = NREG - 1;
  const int32_t entry = stripe_claim(order, claim);
  const PlanTask tk = plan[entry & 0xffffff];
  const int lane = threadIdx.x;
  const int sb = (int)((uint32_t)entry >> 24);
  const int qlen = tk.qlen, tlen = tk.tlen, w = tk.w;
  const BStripeGeom g = bstripe_geom(qlen, tlen, w, NREG);
  if (sb >= g.nst) return;  // (a padding entry of the launch order)
  // (the launch ends with its longest chain of rows: only the tasks that can be that chain issue first.  Measured on
  // the mm8-like batch of 3,000 tasks: top level from 16,384 rows + columns 20.1-20.5 ms, from 28,000 19.5-19.8 ms;
  // a third level below 12,000 the same 19.6-19.9 ms)
  const bool very_long = qlen + tlen >= 28000;
  if (very_long) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(2);
  const int T0 = sb * NSLOT, T1 = T0 + NSLOT;
  const bool has_left = sb > 0, has_right = sb + 1 < g.nst;
  const int nrow = qlen + tlen - 1;
  uint8_t *gsync = dirbase + tk.dir_off + (int64_t)g.nst * (int64_t)g.flag_bytes;
  BStripeRec *recs = reinterpret_cast<BStripeRec *>(gsync);
  // (64-bit words: x | v << 16 | tag in the low half, H in the high half -- one store, one load)
  unsigned long long *cols = reinterpret_cast<unsigned long long *>(gsync + (size_t)g.nst * 64);
  unsigned long long *col_out = cols + (size_t)sb * g.col_len;  // [row - (first row of the right stripe - 1)]
  unsigned long long *col_in = cols + (size_t)(has_left ? sb - 1 : 0) * g.col_len;  // [row - (my first row - 1)]: state AFTER that row
  const int r_a = bstripe_first_row(T0, w);
  int r_z = bstripe_last_row(T1, qlen, tlen, w);
  const int next_a = bstripe_first_row(T1, w);  // the right stripe's first row: it reads my state from row next_a - 1 on
  // rows on which the left stripe's last column is read: while my first 16-cell block is computed
  const int r_need = bstripe_last_row(T0 + 16, qlen, tlen, w);
  const bool with_dir = !(tk.flag & SDF_FLAG_SCORE_ONLY);
  uint4 *dir = reinterpret_cast<uint4 *>(dirbase + tk.dir_off + (int64_t)sb * (int64_t)g.flag_bytes);
  // H of every band cell and the best cell are what a band that runs out needs (the traceback starts at the best
  // cell).  A band that reaches the corner needs score and mte only: H of the top cell and of the one under it, i.e. of
  // this stripe's columns while the top cell is in them or has just left for the right neighbour's first column.
  Band b_end;
  const bool can_drop = !band_of(nrow - 1, qlen, tlen, w, b_end);

  // ---- the stripe's target codes; the reversed query of its rows: W[i] = (QR[i + q0], QR[i + q0 + 1]),
  // QR[e] = query[qlen - 1 - e] (0 outside) -- row r, lane l, register k read e = qlen - 1 - r + T0 + 128 k + 2 l ----
  const int q0 = qlen - 1 - r_z + T0 - 2;  // first entry any row of the stripe reads (minus the fetch-ahead)
  uint16_t *W = reinterpret_cast<uint16_t *>(lds);
  bool has_n;
  unsigned TA[NREG], TB[NREG];  // score tables of the lane's two target positions per register
  const unsigned t_mis4 = (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff) * 0x01010101u, t_wild4 = (unsigned)sc.qe2_b * 0x01010101u;
  const unsigned t_delta = (unsigned)((sc.sc_match + sc.qe2_b) & 0xff) ^ (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff);
  {
    const uint32_t *tw = pool + tk.t_word, *tn = tw + (tlen + 15) / 16;
    const uint32_t *qw = pool + tk.q_word, *qn = qw + (qlen + 15) / 16;
    const int wcap = (r_z - r_a) + NSLOT + 8;
    uint32_t n_seen = 0;
    for (int i = lane; i < wcap; i += 64) {
      const int e0 = i + q0, e1 = e0 + 1;
      uint32_t v0 = (e0 >= 0 && e0 < qlen) ? pool_code16(qw, qn, qlen - 1 - e0, sc.wild) : 0u;
      uint32_t v1 = (e1 >= 0 && e1 < qlen) ? pool_code16(qw, qn, qlen - 1 - e1, sc.wild) : 0u;
      n_seen |= (v0 | v1) >> 8;
      W[i] = qsel_pair(v0, v1);  // (selector form: extz2_wave.hip, SDF_SCORE2)
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int t = T0 + 128 * k + 2 * lane;
      const uint32_t c0 = t < tlen ? pool_code16(tw, tn, t, sc.wild) : 0u;
      const uint32_t c1 = t + 1 < tlen ? pool_code16(tw, tn, t + 1, sc.wild) : 0u;
      n_seen |= (c0 | c1) >> 8;
      TA[k] = score_table(c0, t_mis4, t_delta, t_wild4);
      TB[k] = score_table(c1, t_mis4, t_delta, t_wild4);
    }
    has_n = __builtin_amdgcn_readfirstlane((int)__any(n_seen != 0)) != 0;  // wave-uniform
  }
  __syncthreads();

  // ---- constants of the <<8 difference domain ----
  const unsigned qv = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned capv = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned z_wild = ((unsigned)sc.qe2_b << 8) * 0x00010001u;  // score 0
  unsigned one2 = 0x00010001u;
  SDF_OPQ(one2);

  unsigned U[NREG], V[NREG], X[NREG], Y[NREG], S[NREG];
  unsigned Fa[NREG], Fb[NREG], Fx[NREG], Fy[NREG];
  int32_t He[NREG], Ho[NREG];                          // H of the even / odd column of the lane
  int32_t bHe[NREG], bRe[NREG], bHo[NREG], bRo[NREG];  // best H of the column so far, and its row
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
    U[k] = V[k] = X[k] = Y[k] = 0u;
    S[k] = z_wild;  // (the reference's s array is zero: z = 0 + 2 (q + e))
    Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    He[k] = Ho[k] = SDF_NEG_INF;
    bHe[k] = bHo[k] = 0;  // (ez->max starts at 0: only a positive H can become the maximum, :41)
    bRe[k] = bRo[k] = -1;
  }
  int32_t ez_score = SDF_NEG_INF, ez_mte = SDF_NEG_INF, ez_mte_q = -1;
  bool have_score = false;

  uint32_t feed_xv = 0u, feed_h = 0u, next_xv = 0u, next_h = 0u;
  uint32_t feed_c = 0u;  // (feed_xv without its tag bit: x | v << 16 as the rows take it)
  int feed_r0 = -0x40000000;
  uint32_t out_xv = 0u, out_h = 0u;  // my last column's words of the block (lane = row & 15)
  auto feed_load = [&](const int rfirst, uint32_t &xv, uint32_t &hh) {  // states after rows rfirst - 1 + (0 .. 15)
    const int r = rfirst + (lane & 15);
    xv = 1u;
    hh = 0u;
    if (r >= r_a && r <= r_need) {
      const unsigned long long wv = ld_agent(col_in + (r - r_a));
      xv = (uint32_t)wv;
      hh = (uint32_t)(wv >> 32);
    }
  };

#ifdef SDF_STRIPE_TIMING
  const unsigned long long tm_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tm_first = 0, tm_wait = 0;
  int n_full = 0, n_top_rows = 0, n_lean = 0, n_slow = 0;  // rows by flavour, and the time spent in each
  unsigned long long tm_full = 0, tm_top = 0, tm_lean = 0, tm_slow = 0, tm_r256 = 0;
#endif
  int r_stop = r_z + 1;  // (lowered when the band runs out)
  for (int r0 = r_a & ~15; r0 < r_stop; r0 += 16) {
    r0 = __builtin_amdgcn_readfirstlane(r0);
    const int rb = r0 > r_a ? r0 : r_a, re = r0 + 16 < r_stop ? r0 + 16 : r_stop;
    if (has_left && rb <= r_need) {
      uint32_t gx, gh;
      if (feed_r0 + 16 == r0) {
        gx = next_xv;
        gh = next_h;
      } else {
        feed_load(r0, gx, gh);
      }
      int spins = 0;
#ifdef SDF_STRIPE_TIMING
      const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
      // (before its first block a stripe may wait for most of the task: it looks every few microseconds, so that the
      // hundred waiting stripes of a long task leave the L2 to the ten working ones; between blocks it looks at once)
      const bool first_wait = feed_r0 < 0;
      if (__builtin_amdgcn_readfirstlane((int)__any((gx & 1u) == 0u))) {
        __builtin_amdgcn_s_setprio(0);  // (a waiting wavefront must not take issue slots from the one it waits for)
        do {
          if (first_wait) __builtin_amdgcn_s_sleep(SDF_BS_FIRST_SLEEP);
          else __builtin_amdgcn_s_sleep(2);
          feed_load(r0, gx, gh);
          if (!__builtin_amdgcn_readfirstlane((int)__any((gx & 1u) == 0u))) break;
          // (gives up like the full-band stripe kernel: extz2_stripe.hip, stripe_abandon)
          if (++spins >= spin_cap || ((spins & 63) == 63 && stripe_abandoned(res + tk.out_idx))) {
            stripe_abandon(gave_up, res + tk.out_idx, tk.out_idx, lane);
            return;
          }
        } while (true);
        if (very_long) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(2);
      }
#ifdef SDF_STRIPE_TIMING
      if (!tm_first) tm_first = __builtin_amdgcn_s_memrealtime();
      else if (spins) tm_wait += __builtin_amdgcn_s_memrealtime() - tw0;
#endif
      feed_xv = gx;
      feed_h = gh;
      feed_c = gx & ~1u;
      feed_r0 = r0;
      feed_load(r0 + 16, next_xv, next_h);
    } else {
      feed_xv = 1u;
      feed_h = 0u;
      feed_c = 0u;
    }
    unsigned qaddr = (unsigned)(2 * (qlen - 1 - rb + T0 + 2 * lane - q0));
    unsigned qnext[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
    int r = rb;
    // ---- a block on whose rows the band covers the whole stripe with room to spare (its start left of the stripe,
    // its top cell right of it: most rows of a wide band): every cell is computed, refreshed and an inner cell of the
    // band -- no border cell, no first-cell rule, no masks ----
    {
      Band b_first, b_last;
      // (what is cheap and necessary first -- hi0(rb) <= (rb + w) >> 1, lo0(re - 1) >= (re - w) >> 1: a stripe at an edge of
      // the band, whose blocks are the task's chain of rows, fails here and is spared the two band evaluations)
      bool full = has_left && re > rb && rb >= next_a - 1 && ((rb + w) >> 1) >= T1 && ((re - w) >> 1) < T0;
      if (full) {
        const bool ok_first = band_of(rb, qlen, tlen, w, b_first), ok_last = band_of(re - 1, qlen, tlen, w, b_last);
        full = ok_first && ok_last && b_last.lo0 < T0 && b_first.hi0 >= T1;
      }
#ifdef SDF_BS_NO_FULL
      if (false) {
#else
      if (full) {
#endif
        const bool full_h = can_drop || b_first.hi0 <= T1 + 1;  // (see can_drop)
#ifdef SDF_STRIPE_TIMING
        const unsigned long long tf0 = __builtin_amdgcn_s_memrealtime();
        n_full += re - r;
#endif
#pragma unroll 1
        for (; r < re; ++r) {
          unsigned qc[NREG];
#pragma unroll
          for (int k = 0; k < NREG; ++k) qc[k] = qsel_spread(qnext[k]);
          qaddr -= 2;
          asm volatile("" : "+v"(qaddr) : "v"(qc[0]), "v"(qc[KT]));
#pragma unroll
          for (int k = 0; k < NREG; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
          const int fc = __builtin_amdgcn_readlane((int)feed_c, r - r0);
          unsigned xt1[NREG], vt1[NREG];
          unsigned P[NREG];  // (x and v of the lane's odd column in one word: see the ordinary rows below)
#pragma unroll
          for (int k = 0; k < NREG; ++k) P[k] = __builtin_amdgcn_perm(V[k], X[k], 0x07060302u);
#pragma unroll
          for (int k = 0; k < NREG; ++k) {
            unsigned ps;
            if (k == 0) {
              ps = (unsigned)__builtin_amdgcn_update_dpp(fc, (int)P[0], 0x138, 0xf, 0xf, false);
            } else {
              int up;
              asm("" : "=v"(up));
              const int p0 = __builtin_amdgcn_update_dpp(up, (int)P[k > 0 ? k - 1 : 0], 0x13C, 0x1, 0x1, false);
              ps = (unsigned)__builtin_amdgcn_update_dpp(p0, (int)P[k], 0x138, 0xf, 0xf, false);
            }
            xt1[k] = __builtin_amdgcn_perm(X[k], ps, 0x05040100u);
            vt1[k] = __builtin_amdgcn_perm(V[k], ps, 0x05040302u);
          }
#pragma unroll
          for (int k = 0; k < NREG; ++k) SDF_SCORE2(S[k], k, qc[k], false)
          if (has_n) {  // (an N in the query: the selector picked 0xff)
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
              unsigned nn = pk_ashr15(S[k]);
              SDF_OPQ(nn);
              S[k] = (z_wild & nn) | (S[k] & ~nn);
            }
          }
#pragma unroll
          for (int k = 0; k < NREG; ++k) SDF_CORE(k)
          if (full_h) {
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
              const int32_t nE = He[k] + (int32_t)((V[k] >> 8) & 0xffu) - sc.qe;
              const int32_t nO = Ho[k] + (int32_t)(V[k] >> 24) - sc.qe;
              He[k] = nE;
              Ho[k] = nO;
              if (can_drop) {
                const bool gE = nE > bHe[k], gO = nO > bHo[k];
                bHe[k] = gE ? nE : bHe[k];
                bRe[k] = gE ? r : bRe[k];
                bHo[k] = gO ? nO : bHo[k];
                bRo[k] = gO ? r : bRo[k];
              }
            }
          }
          if (has_right) {
            const unsigned ew = __builtin_amdgcn_perm(V[KT], X[KT], 0x07060302u);
            const unsigned es = (unsigned)__builtin_amdgcn_readlane((int)ew, 63) | 1u;
            const unsigned eh = (unsigned)__builtin_amdgcn_readlane(Ho[KT], 63);
            out_xv = lane == (r & 15) ? es : out_xv;
            out_h = lane == (r & 15) ? eh : out_h;
          }
        }
#ifdef SDF_STRIPE_TIMING
        tm_full += __builtin_amdgcn_s_memrealtime() - tf0;
#endif
      }
    }
    // ---- ordinary rows, as long as they last (no border cell: hi < r): lane predicates against the row's band; the
    // row on which the first computed cell's window moves (one in 32) takes the one branch ----
    // The row's scalars -- band ends, computed blocks, the ranges of the score refresh -- are a dozen shifts, masks and
    // compares each on the scalar unit; a stripe at the edge of the band is ONE wavefront on its SIMD, every instruction of
    // it an issue slot of the task's chain of rows (round 4: 204 instructions a row, 103 of them scalar).  They depend on
    // the row number alone: lane i of a few registers computes them for row r0 + i once per 16-row block, and a row reads
    // its own with one v_readlane each.
    int t_lo = 0, t_fix = 0, t_act = 0, t_a1 = 0, t_l1 = 0, t_a0 = 0, t_l0 = 0, t_lo0 = 0, t_span = 0;
    int t_actn = 0, t_b1 = 0, t_b0 = 0;  // (the same ranges as bounds from lane 0, for rows whose band starts left of the stripe)
    unsigned m_stop = 0xffffu, m_moved = 0u, m_hrow = 0u, m_end = 0u, m_top = 0u;
    if (r < re) {  // (a block of rows on which the band covers the whole stripe is done by now)
      const int rr = r0 + (lane & 15);
      int c_lo0 = (rr - w + 1) >> 1, c_hi0 = (rr + w) >> 1, c_plo = (rr - w) >> 1;
      c_lo0 = c_lo0 < rr - qlen + 1 ? rr - qlen + 1 : c_lo0;
      c_lo0 = c_lo0 < 0 ? 0 : c_lo0;
      c_hi0 = c_hi0 > rr ? rr : c_hi0;
      c_hi0 = c_hi0 > tlen - 1 ? tlen - 1 : c_hi0;
      c_plo = c_plo < rr - qlen ? rr - qlen : c_plo;
      c_plo = c_plo < 0 ? 0 : c_plo;
      const int c_lo = c_lo0 & ~15, c_hi = c_hi0 | 15, c_prev = c_plo & ~15;
      const bool c_moved = c_lo != c_prev && c_lo >= T0 && c_lo < T1;
      const bool c_stop = c_lo0 > c_hi0 || rr == 0 || c_hi >= rr || (c_lo == 0 && T0 == 0);  // (not an ordinary row)
      const int c_ra = c_lo0 - T0, c_rbe = c_ra + ((c_hi0 - c_lo0) & ~15) + 16;
      const int c_a1 = (c_ra + 1) >> 1, c_b1 = (c_rbe + 1) >> 1, c_a0 = c_ra >> 1, c_b0 = c_rbe >> 1;
      t_lo = c_lo;
      t_lo0 = c_lo0;
      t_fix = c_lo > 0 && !c_moved ? c_lo : -2;
      t_a1 = c_a1;
      t_a0 = c_a0;
      t_l1 = c_b1 > c_a1 ? c_b1 - c_a1 : 0;
      t_l0 = c_b0 > c_a0 ? c_b0 - c_a0 : 0;
      t_act = c_hi - c_lo;
      t_span = c_hi0 - c_lo0;
      t_actn = (c_hi + 1 - T0) >> 1;
      t_b1 = c_b1;
      t_b0 = c_b0;
      m_top = (unsigned)__ballot(c_lo0 < T0) & 0xffffu;
      m_stop = (unsigned)__ballot(c_stop) & 0xffffu;
      m_moved = (unsigned)__ballot(c_moved) & 0xffffu;
      m_hrow = can_drop ? 0xffffu : (unsigned)__ballot(c_hi0 <= T1 + 1) & 0xffffu;  // (see can_drop)
      m_end = (unsigned)__ballot(c_hi0 == tlen - 1 && c_hi0 >= T0 && c_hi0 < T1) & 0xffffu;
    }
#pragma unroll 1
    for (; r < re; ++r) {
#ifndef SDF_BS_NO_LEAN
      {
        // (the run of ordinary rows ahead: up to the next row the table marks, or the end of the block)
        r = __builtin_amdgcn_readfirstlane(r);
        const unsigned ahead = m_stop >> (r - r0);
        int r_end = ahead ? r + __builtin_ctz(ahead) : re;
        r_end = r_end < re ? r_end : re;
        const int out_from = has_right ? next_a - 1 : 0x7fffffff;
        // TOP: the band starts left of the stripe on every row of the run (the stripe holds the band's upper edge: the rows
        // of a long task's chain that no other stripe can overlap) -- no first computed cell, no moved window, and every
        // range starts at lane 0: one bound and one compare each
        const unsigned not_top = (~m_top & 0xffffu) >> (r - r0);
        const int n_top = not_top ? __builtin_ctz(not_top) : 16;
        if (n_top > 0 && r + n_top < r_end) r_end = r + n_top;  // (the rest of the rows: the next turn of the outer loop)
        auto lean_run = [&](auto top_c) {
        constexpr bool TOP = decltype(top_c)::value;
#pragma unroll 1
        for (int rl = r; rl < r_end; ++rl) {
          const int ri = rl - r0;
          const int lo = TOP ? 0 : __builtin_amdgcn_readlane(t_lo, ri), lo_fix = TOP ? 0 : __builtin_amdgcn_readlane(t_fix, ri);
          const int act_span = TOP ? 0 : __builtin_amdgcn_readlane(t_act, ri);
          const int a1 = TOP ? 0 : __builtin_amdgcn_readlane(t_a1, ri), l1 = TOP ? 0 : __builtin_amdgcn_readlane(t_l1, ri);
          const int a0 = TOP ? 0 : __builtin_amdgcn_readlane(t_a0, ri), l0 = TOP ? 0 : __builtin_amdgcn_readlane(t_l0, ri);
          const int actn = TOP ? __builtin_amdgcn_readlane(t_actn, ri) : 0;
          const int b1 = TOP ? __builtin_amdgcn_readlane(t_b1, ri) : 0, b0 = TOP ? __builtin_amdgcn_readlane(t_b0, ri) : 0;
          const int lo0 = __builtin_amdgcn_readlane(t_lo0, ri);
          const unsigned span = (unsigned)__builtin_amdgcn_readlane(t_span, ri);
          unsigned qc[NREG];
#pragma unroll
          for (int k = 0; k < NREG; ++k) qc[k] = qsel_spread(qnext[k]);
          qaddr -= 2;
          asm volatile("" : "+v"(qaddr) : "v"(qc[0]), "v"(qc[KT]));
#pragma unroll
          for (int k = 0; k < NREG; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
          // The (r-1, t-1) neighbours: the odd column of the lane to the left.  x and v of that column travel TOGETHER -- the
          // word the stripes hand each other (x | v << 16, feed_c: the tag bit cleared) is also what one DPP shift moves
          const int fc = __builtin_amdgcn_readlane((int)feed_c, ri);
          const int32_t fh = __builtin_amdgcn_readlane((int)feed_h, ri);
          unsigned xt1[NREG], vt1[NREG];
          int32_t hleft[NREG];  // H (before this row) of the column to the left of the lane's even one
          unsigned P[NREG];
#pragma unroll
          for (int k = 0; k < NREG; ++k) P[k] = __builtin_amdgcn_perm(V[k], X[k], 0x07060302u);
#pragma unroll
          for (int k = 0; k < NREG; ++k) {
            unsigned ps;
            if (k == 0) {
              ps = (unsigned)__builtin_amdgcn_update_dpp(fc, (int)P[0], 0x138, 0xf, 0xf, false);
              hleft[0] = __builtin_amdgcn_update_dpp(fh, Ho[0], 0x138, 0xf, 0xf, false);
            } else {
              int up, uh;
              asm("" : "=v"(up));
              asm("" : "=v"(uh));
              const int p0 = __builtin_amdgcn_update_dpp(up, (int)P[k > 0 ? k - 1 : 0], 0x13C, 0x1, 0x1, false);
              ps = (unsigned)__builtin_amdgcn_update_dpp(p0, (int)P[k], 0x138, 0xf, 0xf, false);
              const int h0 = __builtin_amdgcn_update_dpp(uh, Ho[k > 0 ? k - 1 : 0], 0x13C, 0x1, 0x1, false);
              hleft[k] = __builtin_amdgcn_update_dpp(h0, Ho[k], 0x138, 0xf, 0xf, false);
            }
            xt1[k] = __builtin_amdgcn_perm(X[k], ps, 0x05040100u);  // (neighbour's x | my even column's x << 16)
            vt1[k] = __builtin_amdgcn_perm(V[k], ps, 0x05040302u);
          }
          // the first computed cell (lo: an even column), its window not moved: its left neighbour reads as 0 (lo_fix)
          if (!TOP && ((m_moved >> ri) & 1u)) {  // moved (one row in 32): the slot to the left as it is, and a negative byte there also
                                       // sets the next three cells (the reference's sign extension, :145-146)
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
              const int tb = T0 + 128 * k;
              if (lo < tb || lo > tb + 127) continue;
              const int ll = (lo - tb) >> 1;
              const unsigned cx = (unsigned)__builtin_amdgcn_readlane((int)xt1[k], ll) & 0xffffu;
              const unsigned cv = (unsigned)__builtin_amdgcn_readlane((int)vt1[k], ll) & 0xffffu;
              const unsigned sx = (cx & 0x8000u) ? 0xff00u : 0u, sv = (cv & 0x8000u) ? 0xff00u : 0u;
              const unsigned mx = lane == ll ? sx << 16 : lane == ll + 1 ? sx * 0x00010001u : 0u;
              const unsigned mv = lane == ll ? sv << 16 : lane == ll + 1 ? sv * 0x00010001u : 0u;
              xt1[k] |= mx;
              vt1[k] |= mv;
            }
          }
          const bool h_row = (m_hrow >> ri) & 1u;
#pragma unroll
          for (int k = 0; k < NREG; ++k) {
            const int te = T0 + 128 * k + 2 * lane;
            if (!TOP) {
              unsigned keep = te == lo_fix ? 0xffff0000u : 0xffffffffu;
              SDF_OPQ(keep);
              xt1[k] &= keep;
              vt1[k] &= keep;
            }
            unsigned z;
            SDF_SCORE2(z, k, qc[k], has_n)
            if (TOP) {
              sel_lo_below(S[k], z, b1 - 64 * k, lane);
              sel_hi_below(S[k], z, b0 - 64 * k, lane);
            } else {
              sel_lo_len(S[k], z, a1 - 64 * k, l1, lane);
              sel_hi_len(S[k], z, a0 - 64 * k, l0, lane);
            }
            const bool act = TOP ? lane < actn - 64 * k : (unsigned)(te - lo) <= (unsigned)act_span;
            const unsigned a_ = pk_add(xt1[k], vt1[k]);
            const unsigned bb_ = pk_add(Y[k], U[k]);
            const unsigned z0_ = S[k];
            const unsigned z1_ = pk_maxi(z0_, a_);
            const unsigned fa_ = z1_ - z0_;  // (32-bit: never borrows, see SDF_CORE)
            const unsigned zb_ = pk_maxi(z1_, bb_);
            const unsigned fb_ = zb_ - z1_;
            const unsigned z2_ = pk_maxu(z1_, bb_);
            const unsigned z3_ = pk_minu(z2_, capv);
            const unsigned un_ = pk_sub(z3_, vt1[k]);
            const unsigned vn_ = pk_sub(z3_, U[k]);
            const unsigned zq_ = z3_ - qv;
            const unsigned xn_ = pk_maxi(pk_sub(a_, zq_), 0u);
            const unsigned yn_ = pk_maxi(pk_sub(bb_, zq_), 0u);
            U[k] = act ? un_ : U[k];
            V[k] = act ? vn_ : V[k];
            X[k] = act ? xn_ : X[k];
            Y[k] = act ? yn_ : Y[k];
            Fa[k] = shl1_or(Fa[k], pk_nonzero(fa_));
            Fb[k] = shl1_or(Fb[k], pk_nonzero(fb_));
            Fx[k] = shl1_or(Fx[k], pk_nonzero(xn_));
            Fy[k] = shl1_or(Fy[k], pk_nonzero(yn_));
            // H: the top cell (hi0 > 0 here) from the column to its left before this row, the cells below it from
            // themselves
            if (!h_row) continue;  // (wave-uniform; see can_drop)
            const int32_t vE = (int32_t)((V[k] >> 8) & 0xffu), vO = (int32_t)(V[k] >> 24);
            const int32_t uE = (int32_t)((U[k] >> 8) & 0xffu), uO = (int32_t)(U[k] >> 24);
            const unsigned dE = (unsigned)(te - lo0), dO = dE + 1u;
            // (every candidate computed on all lanes, then two selects: the compiler would otherwise mask EXEC around each)
            int32_t tE = hleft[k] + uE - sc.qe, tO = He[k] + uO - sc.qe;
            int32_t mE = He[k] + vE - sc.qe, mO = Ho[k] + vO - sc.qe;
            SDF_OPQ(tE);
            SDF_OPQ(tO);
            SDF_OPQ(mE);
            SDF_OPQ(mO);
            int32_t nE = dE == span ? tE : He[k], nO = dO == span ? tO : Ho[k];
            nE = dE < span ? mE : nE;
            nO = dO < span ? mO : nO;
            He[k] = nE;
            Ho[k] = nO;
            if (can_drop) {
              const bool gE = dE <= span && nE > bHe[k], gO = dO <= span && nO > bHo[k];
              bHe[k] = gE ? nE : bHe[k];
              bRe[k] = gE ? rl : bRe[k];
              bHo[k] = gO ? nO : bHo[k];
              bRo[k] = gO ? rl : bRo[k];
            }
          }
          if ((m_end >> ri) & 1u) {  // the end of the target: mte, score (:259-262)
            const int hi0 = tlen - 1, hi = hi0 | 15;
            const int st = hi0 - T0;
            int32_t hv = 0;
#pragma unroll
            for (int k = 0; k < NREG; ++k)
              if ((st >> 7) == k) hv = (st & 1) ? __builtin_amdgcn_readlane(Ho[k], (st & 127) >> 1) : __builtin_amdgcn_readlane(He[k], (st & 127) >> 1);
            if (hv > ez_mte) {
              ez_mte = hv;
              ez_mte_q = rl - hi;
            }
            if (rl == nrow - 1) {
              ez_score = hv;
              have_score = true;
            }
          }
          if (rl >= out_from) {  // my last column after this row, for the right stripe: lane row & 15 of the two words
            const unsigned ew = __builtin_amdgcn_perm(V[KT], X[KT], 0x07060302u);
            const unsigned es = (unsigned)__builtin_amdgcn_readlane((int)ew, 63) | 1u;
            const unsigned eh = (unsigned)__builtin_amdgcn_readlane(Ho[KT], 63);
            // (r0 is a multiple of 16: lane rl & 15 is lane ri; the lane select of v_writelane next to an SGPR value is M0,
            // which nothing else in this kernel uses)
            uint32_t ox = out_xv, oh = out_h;  // (locals: the operands of an asm statement inside a generic lambda)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0"
                         : "+v"(ox), "+v"(oh) : "s"(ri), "s"(es), "s"(eh) : "m0");
#pragma clang diagnostic pop
            out_xv = ox;
            out_h = oh;
          }
        }
        };
#ifdef SDF_STRIPE_TIMING
        const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
        if (n_top > 0) n_top_rows += r_end - r;
        else n_lean += r_end - r;
#endif
        if (n_top > 0) lean_run(std::true_type{});
        else lean_run(std::false_type{});
#ifdef SDF_STRIPE_TIMING
        if (n_top > 0) tm_top += __builtin_amdgcn_s_memrealtime() - tl0;
        else tm_lean += __builtin_amdgcn_s_memrealtime() - tl0;
        if (r <= r_a + 256 && r_end > r_a + 256) tm_r256 = __builtin_amdgcn_s_memrealtime();
#endif
        r = r_end;
      }
#endif
      if (r >= re) break;
#ifdef SDF_STRIPE_TIMING
      const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
      ++n_slow;
#endif
      Band bd, bp;
      if (!band_of(r, qlen, tlen, w, bd)) {  // the band has run out (every stripe sees it on the same row)
        r_stop = r;
        break;
      }
      band_of(r - 1, qlen, tlen, w, bp);
      const int lo0 = bd.lo0, hi0 = bd.hi0, lo = bd.lo, hi = bd.hi;
      const int prev_lo = r > 0 ? bp.lo : -1;
      unsigned qc[NREG];
#pragma unroll
      for (int k = 0; k < NREG; ++k) qc[k] = qsel_spread(qnext[k]);
      qaddr -= 2;
      asm volatile("" : "+v"(qaddr) : "v"(qc[0]), "v"(qc[KT]));
#pragma unroll
      for (int k = 0; k < NREG; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
      // the left column after row r - 1
      const uint32_t fxv = (uint32_t)__builtin_amdgcn_readlane((int)feed_xv, r - r0) & ~1u;
      const int32_t fh = __builtin_amdgcn_readlane((int)feed_h, r - r0);
      // ---- border cell t = r (:122), inside the computed blocks only ----
      if (hi >= r && r >= T0 && r < T1) {
        const int sr = r - T0;
        const unsigned keep = (sr & 1) ? 0x0000ffffu : 0xffff0000u;
        const unsigned uval = r ? (((unsigned)sc.q_b << 8) << ((sr & 1) * 16)) : 0u;
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          const bool mine = (sr >> 7) == k && lane == ((sr & 127) >> 1);
          U[k] = mine ? ((U[k] & keep) | uval) : U[k];
          Y[k] = mine ? (Y[k] & keep) : Y[k];
        }
      }
      // ---- per register: neighbours, scores, recurrence on the computed blocks, H ----
      int32_t h_left = fh;          // H (before this row) of the column left of the register's first one
      unsigned xprev = 0u, vprev = 0u;  // x, v of the register to the left as the previous row left them
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        const int tb = T0 + 128 * k;  // first column of the register
        const int32_t h_left_k = h_left;
        const unsigned xleft = xprev, vleft = vprev;
        h_left = __builtin_amdgcn_readlane(Ho[k], 63);
        xprev = X[k];
        vprev = V[k];
        if (hi < tb || lo > tb + 127) {  // (wave-uniform) no computed cell in this register: its flag rows move on
          Fa[k] <<= 1;
          Fb[k] <<= 1;
          Fx[k] <<= 1;
          Fy[k] <<= 1;
          continue;
        }
        const int te = tb + 2 * lane;
        // (r-1, t-1) neighbours
        unsigned xs, vs;
        if (k == 0) {
          unsigned xc = fxv << 16, vc = fxv & 0xffff0000u;
          if (!has_left) {
            xc = 0u;
            vc = r ? ((unsigned)sc.q_b << 24) : 0u;
          }
          xs = (unsigned)__builtin_amdgcn_update_dpp((int)xc, (int)X[0], 0x138, 0xf, 0xf, false);
          vs = (unsigned)__builtin_amdgcn_update_dpp((int)vc, (int)V[0], 0x138, 0xf, 0xf, false);
        } else {
          int ux, uv;
          asm("" : "=v"(ux));
          asm("" : "=v"(uv));
          const int x0 = __builtin_amdgcn_update_dpp(ux, (int)xleft, 0x13C, 0x1, 0x1, false);
          xs = (unsigned)__builtin_amdgcn_update_dpp(x0, (int)X[k], 0x138, 0xf, 0xf, false);
          const int v0 = __builtin_amdgcn_update_dpp(uv, (int)vleft, 0x13C, 0x1, 0x1, false);
          vs = (unsigned)__builtin_amdgcn_update_dpp(v0, (int)V[k], 0x138, 0xf, 0xf, false);
        }
        unsigned xt1 = __builtin_amdgcn_alignbit(X[k], xs, 16);
        unsigned vt1 = __builtin_amdgcn_alignbit(V[k], vs, 16);
        // the first computed cell (:140-146)
        if (lo >= tb && lo <= tb + 127 && lo > 0) {
          const int ll = (lo - tb) >> 1;  // its lane (lo is even: the low half)
          if (lo == prev_lo) {            // the window did not move: that neighbour reads as 0
            if (lane == ll) {
              xt1 &= 0xffff0000u;
              vt1 &= 0xffff0000u;
            }
          } else {  // it moved: the slot to the left, and a negative byte also sets the next three cells (sign extension)
            const unsigned cx = (unsigned)__builtin_amdgcn_readlane((int)xt1, ll) & 0xffffu;
            const unsigned cv = (unsigned)__builtin_amdgcn_readlane((int)vt1, ll) & 0xffffu;
            const unsigned sx = (cx & 0x8000u) ? 0xff00u : 0u, sv = (cv & 0x8000u) ? 0xff00u : 0u;
            if (lane == ll) {
              xt1 |= sx << 16;
              vt1 |= sv << 16;
            }
            if (lane == ll + 1) {
              xt1 |= sx * 0x00010001u;
              vt1 |= sv * 0x00010001u;
            }
          }
        }
        // scores: refreshed in 16-cell strides from lo0 (:124-138)
        {
          unsigned z;
          SDF_SCORE2(z, k, qc[k], has_n)
          const int ra = lo0 - tb, rbe = ra + ((hi0 - lo0) & ~15) + 16;
          sel_lo_rng(S[k], z, (ra + 1) >> 1, (rbe + 1) >> 1, lane);
          sel_hi_rng(S[k], z, ra >> 1, rbe >> 1, lane);
        }
        // recurrence; written back in the computed blocks [lo, hi] only
        const bool act = (unsigned)(te - lo) <= (unsigned)(hi - lo);
        {
          const unsigned a_ = pk_add(xt1, vt1);
          const unsigned bb_ = pk_add(Y[k], U[k]);
          const unsigned z0_ = S[k];
          const unsigned z1_ = pk_maxi(z0_, a_);
          const unsigned fa_ = z1_ - z0_;  // (32-bit: never borrows, see SDF_CORE)
          const unsigned zb_ = pk_maxi(z1_, bb_);
          const unsigned fb_ = zb_ - z1_;
          const unsigned z2_ = pk_maxu(z1_, bb_);
          const unsigned z3_ = pk_minu(z2_, capv);
          const unsigned un_ = pk_sub(z3_, vt1);
          const unsigned vn_ = pk_sub(z3_, U[k]);
          const unsigned zq_ = z3_ - qv;
          const unsigned xn_ = pk_maxi(pk_sub(a_, zq_), 0u);
          const unsigned yn_ = pk_maxi(pk_sub(bb_, zq_), 0u);
          U[k] = act ? un_ : U[k];
          V[k] = act ? vn_ : V[k];
          X[k] = act ? xn_ : X[k];
          Y[k] = act ? yn_ : Y[k];
          Fa[k] = shl1_or(Fa[k], pk_nonzero(fa_));
          Fb[k] = shl1_or(Fb[k], pk_nonzero(fb_));
          Fx[k] = shl1_or(Fx[k], pk_nonzero(xn_));
          Fy[k] = shl1_or(Fy[k], pk_nonzero(yn_));
        }
        // H of the band cells (:222-258): the top cell from the column to its left (before this row), the others
        // from themselves
        {
          const int32_t vE = (int32_t)((V[k] >> 8) & 0xffu), vO = (int32_t)(V[k] >> 24);
          const int32_t uE = (int32_t)((U[k] >> 8) & 0xffu), uO = (int32_t)(U[k] >> 24);
          const int32_t ho_left = __builtin_amdgcn_update_dpp(h_left_k, Ho[k], 0x138, 0xf, 0xf, false);  // H of te - 1
          const bool midE = (unsigned)(te - lo0) < (unsigned)(hi0 - lo0);
          const bool midO = (unsigned)(te + 1 - lo0) < (unsigned)(hi0 - lo0);
          const bool topE = te == hi0, topO = te + 1 == hi0;
          const int32_t tE = r == 0 ? vE - 2 * sc.qe : (hi0 > 0 ? ho_left + uE : He[k] + vE) - sc.qe;
          const int32_t tO = He[k] + uO - sc.qe;  // (hi0 odd: > 0)
          const int32_t nE = topE ? tE : midE ? He[k] + vE - sc.qe : He[k];
          const int32_t nO = topO ? tO : midO ? Ho[k] + vO - sc.qe : Ho[k];
          He[k] = nE;
          Ho[k] = nO;
          const bool gE = (midE || topE) && nE > bHe[k], gO = (midO || topO) && nO > bHo[k];
          bHe[k] = gE ? nE : bHe[k];
          bRe[k] = gE ? r : bRe[k];
          bHo[k] = gO ? nO : bHo[k];
          bRo[k] = gO ? r : bRo[k];
        }
      }
      // ---- the end of the target: mte, score (:259-262) ----
      if (hi0 == tlen - 1 && hi0 >= T0 && hi0 < T1) {
        const int st = hi0 - T0;
        int32_t hv = 0;
#pragma unroll
        for (int k = 0; k < NREG; ++k)
          if ((st >> 7) == k) hv = (st & 1) ? __builtin_amdgcn_readlane(Ho[k], (st & 127) >> 1) : __builtin_amdgcn_readlane(He[k], (st & 127) >> 1);
        if (hv > ez_mte) {
          ez_mte = hv;
          ez_mte_q = r - hi;
        }
        if (r == nrow - 1) {
          ez_score = hv;
          have_score = true;
        }
      }
      // ---- my last column after this row, for the right stripe ----
      if (has_right && r >= next_a - 1) {
        const unsigned ew = __builtin_amdgcn_perm(V[KT], X[KT], 0x07060302u);
        const unsigned es = (unsigned)__builtin_amdgcn_readlane((int)ew, 63) | 1u;
        const unsigned eh = (unsigned)__builtin_amdgcn_readlane(Ho[KT], 63);
        out_xv = lane == (r & 15) ? es : out_xv;
        out_h = lane == (r & 15) ? eh : out_h;
      }
#ifdef SDF_STRIPE_TIMING
      tm_slow += __builtin_amdgcn_s_memrealtime() - ts0;
#endif
    }
    // ---- block end: direction flags and edge words of these rows leave for HBM ----
    const int done_hi = r - 1;  // last row done in this block
    if (with_dir && r > rb) {
      const unsigned sh = 15 - (done_hi & 15);
      const int blk = (r0 >> 4) - (r_a >> 4);
#pragma unroll
      for (int k = 0; k < NREG; ++k)
        dir[((int64_t)blk * NREG + k) * 64 + lane] = make_uint4(pk_shl(Fa[k], sh), pk_shl(Fb[k], sh), pk_shl(Fx[k], sh), pk_shl(Fy[k], sh));
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    if (has_right) {
      const int row = r0 + lane;  // state after `row` -> entry row - (next_a - 1)
      if (lane < 16 && (out_xv & 1u) && row >= next_a - 1 && row - (next_a - 1) < g.col_len)
        st_agent(col_out + (row - (next_a - 1)), (unsigned long long)out_xv | ((unsigned long long)out_h << 32));
      out_xv = 0u;
    }
  }
#ifdef SDF_STRIPE_TIMING
  if (lane == 0 && (sb % 8 == 0 || sb == g.nst - 1))
    printf("bstripe %d rows %d..%d start %llu first %llu r256 %llu end %llu wait %llu | full %d rows %llu, top %d rows %llu, edge %d rows %llu, general %d rows %llu\n",
           sb, r_a, r_stop - 1, tm_start, tm_first, tm_r256, (unsigned long long)__builtin_amdgcn_s_memrealtime(), tm_wait, n_full, tm_full,
           n_top_rows, tm_top, n_lean, tm_lean, n_slow, tm_slow);
#endif
  // the right stripe may read my last column for a few rows after my last one: the final state, repeated
  if (has_right) {
    const int last = r_stop - 1;  // my last row
    const unsigned ew = __builtin_amdgcn_perm(V[KT], X[KT], 0x07060302u);
    const unsigned es = (unsigned)__builtin_amdgcn_readlane((int)ew, 63) | 1u;
    const unsigned eh = (unsigned)__builtin_amdgcn_readlane(Ho[KT], 63);
    const int from = last + 1 > next_a - 1 ? last + 1 : next_a - 1;
    const int e = from + lane - (next_a - 1);
    if (e >= 0 && e < g.col_len) st_agent(col_out + e, (unsigned long long)es | ((unsigned long long)eh << 32));
  }
  // ---- the stripe's record: best cell in the reference's order (:226-258 inside a row, :41 across rows) ----
  BestCell best = {0, -1, 0, -1};
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int rr = half ? bRo[k] : bRe[k];
      if (rr < 0) continue;
      Band br;
      band_of(rr, qlen, tlen, w, br);
      const int t = T0 + 128 * k + 2 * lane + half;
      const int vec_end = br.lo0 + (br.hi0 - br.lo0) / 4 * 4;
      const int key = t == br.hi0 ? 0 : t < vec_end ? 1 + (((t - br.lo0) & 3) << 20) + t : 1 + (4 << 20) + t;
      const BestCell cand = {half ? bHo[k] : bHe[k], rr, key, t};
      if (beats(cand, best)) best = cand;
    }
  }
  best = wave_best(best);
  if (lane == 0) {
    BStripeRec rec;
    rec.bestH = best.H;
    rec.bestR = best.r;
    rec.bestKey = best.key;
    rec.bestT = best.t;
    rec.score = ez_score;
    rec.mte = ez_mte;
    rec.mte_q = ez_mte_q;
    rec.flags = 2 | (have_score ? 1 : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) rec.pad[i] = 0;
    recs[sb] = rec;
  }
}

template __global__ void extz2_bstripe_kernel<1>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                 sdf_result *, unsigned long long *, int, unsigned *);
template __global__ void extz2_bstripe_kernel<2>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                 sdf_result *, unsigned long long *, int, unsigned *);
template __global__ void extz2_bstripe_kernel<4>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                 sdf_result *, unsigned long long *, int, unsigned *);

// Before the launch, one workgroup per launch-order entry (task, stripe): the stripe's record and the edge column of its
// right boundary to zero (no word tagged as written)
__global__ __launch_bounds__(64) void bstripe_init_kernel(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                                          int nreg, uint8_t *__restrict__ dirbase) {
  const int32_t entry = order[blockIdx.x];
  const PlanTask tk = plan[entry & 0xffffff];
  const int sb = (int)((uint32_t)entry >> 24);
  const BStripeGeom g = bstripe_geom(tk.qlen, tk.tlen, tk.w, nreg);
  if (sb >= g.nst) return;
  uint8_t *gsync = dirbase + tk.dir_off + (int64_t)g.nst * (int64_t)g.flag_bytes;
  if (threadIdx.x < 16) reinterpret_cast<int32_t *>(gsync)[sb * 16 + threadIdx.x] = 0;
  if (sb + 1 < g.nst) {
    unsigned long long *col = reinterpret_cast<unsigned long long *>(gsync + (size_t)g.nst * 64) + (size_t)sb * g.col_len;
    for (int i = threadIdx.x; i < g.col_len; i += 64) col[i] = 0ull;
  }
}

// After the launch, one thread per launch-order entry; the thread of a task's stripe 0 merges the stripes' records
__global__ __launch_bounds__(64) void bstripe_finish_kernel(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                                            int n, int nreg, const uint8_t *__restrict__ dirbase,
                                                            sdf_result *__restrict__ res) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= n) return;
  const int32_t entry = order[e];
  if ((uint32_t)entry >> 24) return;
  const PlanTask tk = plan[entry & 0xffffff];
  const BStripeGeom g = bstripe_geom(tk.qlen, tk.tlen, tk.w, nreg);
  const BStripeRec *recs = reinterpret_cast<const BStripeRec *>(dirbase + tk.dir_off + (int64_t)g.nst * (int64_t)g.flag_bytes);
  // did the band run out before the last anti-diagonal?  (its lower bound grows faster than the upper one once they
  // have crossed: empty on some row <=> empty on the last one)
  Band bl;
  const bool dropped = !band_of(tk.qlen + tk.tlen - 2, tk.qlen, tk.tlen, tk.w, bl);
  BestCell best = {0, -1, 0, -1};
  sdf_result o;
  o.score = SDF_NEG_INF;
  o.mte = SDF_NEG_INF;
  o.mte_q = -1;
  for (int s = 0; s < g.nst; ++s) {
    const BStripeRec rc = recs[s];
    if (!(rc.flags & 2)) continue;  // (a stripe the band never reached)
    const BestCell cand = {rc.bestH, rc.bestR, rc.bestKey, rc.bestT};
    if (rc.bestR >= 0 && beats(cand, best)) best = cand;
    if (rc.flags & 1) o.score = rc.score;
    if (rc.mte > o.mte) {
      o.mte = rc.mte;
      o.mte_q = rc.mte_q;
    }
  }
  // (the one-task kernels report the best cell only when the traceback needs it: the band ran out)
  o.max = dropped && best.r >= 0 ? best.H : 0;
  o.max_t = dropped && best.r >= 0 ? best.t : -1;
  o.max_q = dropped && best.r >= 0 ? best.r - best.t : -1;
  o.mqe = SDF_NEG_INF;
  o.mqe_t = -1;
  o.zdropped = dropped ? 1 : 0;
  o.n_cigar = res[tk.out_idx].n_cigar == -1 ? -1 : 0;  // (-1: a stripe gave the task up; it is run again, sdf_launch.hip)
  o.cigar_off = 0;
  o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
  res[tk.out_idx] = o;
}

}  // namespace sdf
