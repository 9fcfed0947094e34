// Register-resident extz2 DP kernel for gfx950: one wavefront per PAIR of DP tasks of equal geometry.
//
// Same results as extz2_wave.hip (and therefore as the reference kernel,
// extern/ksw2_extz2_sse.cc:23-298) for the fields SEDEF consumes -- CIGAR, score, mte.  The band
// schedule of the reference (which cells are computed on which anti-diagonal, where the 16-cell
// blocks start, which scores are refreshed) depends on (qlen, tlen, w) only, so two tasks with the
// same triple share every lane predicate and every scalar decision.  They are packed side by side
// in the 16-bit halves of each register:
//
//   * lane l of register k owns window slot 64k+l of BOTH tasks: task A in the low half, task B
//     in the high half (value<<8 each, as in extz2_wave.hip: the packed 16-bit ALU reproduces the
//     reference's wrap-around int8 arithmetic);
//   * the (r-1, t-1) neighbour is a plain wavefront DPP shift (no half-word realignment), window
//     ranges are whole-lane compares (no SDWA half selects), the per-row scalar work is shared;
//   * a window of `need` slots takes ceil(need/64) registers for two tasks instead of
//     ceil(need/128) per task -- at w=128 (176 slots) 1.5 instead of 2 register-rows per task-row;
//   * direction flags leave as 8 bytes per lane per task per 16 rows (four 16-bit row masks),
//     0.5 B per slot as before.
//
// This file is compiled inside sdf_unity.hip after extz2_wave.hip and uses its helpers
// (pk_*, SDF_OPQ, SDF_CORE).  Special rows (first rows, captured carries, the sign-extension
// artefact of the reference's carry-in) are data dependent: if either task needs the general row,
// both take it -- it is exact for every row.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "sdf_internal.h"

namespace sdf {

// byte code of a base: 0..3, or 0x80|wild for N
__device__ __forceinline__ uint32_t pool_code8(const uint32_t *codes, const uint32_t *nmask, int k,
                                               uint32_t wild) {
  const uint32_t c = (codes[k >> 4] >> ((k & 15) * 2)) & 3u;
  const uint32_t n = (nmask[k >> 5] >> (k & 31)) & 1u;
  return n ? (0x80u | wild) : c;
}

// lanes [a, b) (a >= 0)
__device__ __forceinline__ bool lane_in(int lane, int a, int b) {
  return (unsigned)(lane - a) < (unsigned)(b > a ? b - a : 0);
}

// lanes [a, b) of the wavefront as a scalar mask (0 <= a; b may exceed 64), and a lane predicate from such a mask: the row
// code's lane ranges follow from scalars, so the masks are SALU work and a select is ONE v_cndmask with an SGPR pair --
// no v_cmp on the lane index (round 5: three VOPC fewer per steady row)
__device__ __forceinline__ unsigned long long lane_mask(int a, int b) {
  const unsigned long long hi = b >= 64 ? ~0ull : b <= 0 ? 0ull : (1ull << b) - 1ull;
  const unsigned long long lo = a >= 64 ? ~0ull : a <= 0 ? 0ull : (1ull << a) - 1ull;
  return hi & ~lo;
}
// (never with a compile-time constant: the LOW16 rows' `if (in_mask(0xffffffffffff0000))` computed wrong cells -- the
// compiler folds the builtin on a constant into something else; lane compares stay where the mask is a literal)
__device__ __forceinline__ bool in_mask(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// 0xff00 in the halves whose value is negative (the reference's sign-extended carry, :145-146)
__device__ __forceinline__ unsigned sign_smear(unsigned c) {
  return ((c & 0x8000u) ? 0xff00u : 0u) | ((c & 0x80000000u) ? 0xff000000u : 0u);
}

// entries of the LDS sequence windows: the whole (padded) sequence when it is short, else the window slots plus
// 1024 entries of slack (multiples of 4 keep the query window dword aligned behind the target window)
__host__ __device__ inline int pair_tcap(int tlen, int nreg) {
  const int whole = (tlen + 15) / 16 * 16 + 64 * nreg + 32, win = 64 * nreg + 1024 + 64;
  return whole < win ? whole : win;
}
__host__ __device__ inline int pair_qcap(int qlen, int nreg) {
  const int whole = qlen + 64 * nreg + 36, win = 64 * nreg + 1024 + 68;
  return whole < win ? whole : win;
}

// Fresh (score + 2(q+e)) << 8 of the lane's cell of both tasks: ONE byte permute.  The lane keeps, per window register, a
// table of four score bytes per task -- its target base against query base 0..3 (an N in the target: the wildcard's score
// four times) -- and the query window holds the row's bases as the permute's selector: byte 1 = task A's base (a byte of
// TA), byte 3 = 4 + task B's (a byte of TB), bytes 0 and 2 = 0x0c (zero): the result is 0, zA, 0, zB.  An N in the query
// selects 0xff -- a negative half, patched to the wildcard's score where the sequences hold any N at all (WITH_N).
// (Until the end of round 6: xor, min, multiply-add per cell and row.  The kernel issues a vector instruction on 98 % of
// its SIMD cycles at four cycles each: what counts is their number -- profiles/r06_pair_kernel_pmc.txt.)
#define SDF_PFRESH(z, k, qword, WITH_N)                                 \
  {                                                                     \
    z = __builtin_amdgcn_perm(TB[k], TA[k], (qword));                   \
    if (WITH_N) {                                                       \
      unsigned nn_ = pk_ashr15(z);                                      \
      SDF_OPQ(nn_);                                                     \
      z = (z_wild & nn_) | (z & ~nn_);                                  \
    }                                                                   \
  }
// a query window entry from the two tasks' byte codes (0..3, N: bit 7)
__device__ __forceinline__ uint32_t pair_qsel(uint32_t ca, uint32_t cb) {
  return 0x000c000cu | (((ca & 0x80u) ? 0xffu : ca) << 8) | (((cb & 0x80u) ? 0xffu : cb + 4u) << 24);
}

// Mixed pairs (round 4): the band schedule of the reference, lo0 = max(0, r - qlen + 1, (r - w + 1) >> 1), hi0 = min(tlen - 1, r,
// (r + w) >> 1) (extern/ksw2_extz2_sse.cc:101-115), depends on the lengths only where the r - qlen + 1 / tlen - 1 clips bite:
// on the last ~w anti-diagonals of a task.  Two tasks of the same (w, flag) but different lengths therefore share every lane
// predicate and every scalar decision up to the first row at which a clip bites for either of them.
// pair_clip_free: the last anti-diagonal r such that on ALL rows 0..r neither clip changes the band of a (qlen, tlen, w)
// task AND the top cell is not yet the target's last column (min(r, (r + w) >> 1) < tlen - 1: the row code tests that too).
__host__ __device__ inline int pair_clip_free(int qlen, int tlen, int w) {
  const int a = 2 * qlen - w - 2;  // r - qlen + 1 <= (r - w + 1) >> 1 for every r up to here (and r - qlen + 1 <= 0 while that is negative)
  const int b = tlen - 2 > 2 * tlen - w - 3 ? tlen - 2 : 2 * tlen - w - 3;  // min(r, (r + w) >> 1) < tlen - 1
  return a < b ? a : b;
}
// rows [0, shared) of two tasks with band w run side by side in one wavefront: a multiple of 16 (the kernel works in
// 16-row blocks), and row `shared` itself is still clip-free for both (a row looks one row ahead for its top cell)
__host__ __device__ inline int pair_shared_rows(int qa, int ta, int qb, int tb, int w) {
  const int ca = pair_clip_free(qa, ta, w), cb = pair_clip_free(qb, tb, w);
  const int c = ca < cb ? ca : cb;
  return c >= 16 ? c / 16 * 16 : 0;
}

// STREAM: the sequences do not fit the LDS windows whole (long tasks); without it the window code compiles out.
// TRACK: for banded tasks whose band cannot reach the end of both sequences (|qlen - tlen| > w): the reference runs out
// of band, sets zdropped and backtracks from the best cell seen (extern/ksw2_extz2_sse.cc:116, :292-295), so the exact H
// of EVERY in-band cell is needed, not only along the band's upper edge.  Each lane keeps the H of its cell of task A in
// a 32-bit register (H[t] += v[t] - (q+e); the top cell from its lower neighbour's old H and u, :226-258) and a running
// (best H, row); one reduction at the end applies the reference's tie order (earliest row, then its 4-lane scan order).
// Such a task is launched paired with itself: both halves compute it, half B is ignored.
// MIXED: the two tasks may differ in (qlen, tlen).  Three phases: rows [0, shared) side by side as a virtual task of
// (max qlen, max tlen) -- the shorter query shifted in the LDS window so that one address serves both --, then task A's
// remaining rows with A in both halves (B's state waits in LDS), then task B's with B in both halves: phases two and
// three are the kernel's self-paired mode, entered at row `shared` with the state phase one left.  Tasks of equal
// geometry (or a task paired with itself) run as one phase, as without MIXED.
template <int NREG, bool STREAM, bool TRACK, bool MIXED = false>
__device__ __forceinline__ void pair_body(
    const PlanTask *__restrict__ plan, const int32_t *__restrict__ order, const uint32_t *__restrict__ pool,
    ScoreK sc, uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NSLOT = 64 * NREG;
  const PlanTask tk = plan[order[2 * blockIdx.x]];       // task A (low halves)
  // (a long task is a chain of dependent rows that ends the launch: its wavefront gets the SIMD before those of short
  // tasks sharing it)
  if (tk.qlen + tk.tlen >= 16384) __builtin_amdgcn_s_setprio(3);
  else if (tk.qlen + tk.tlen >= 6144) __builtin_amdgcn_s_setprio(2);
  const PlanTask tkb = plan[order[2 * blockIdx.x + 1]];  // task B (high halves): same qlen, tlen, w, flag
  const int lane = threadIdx.x;
  static_assert(!(MIXED && TRACK), "band-exhausting tasks run paired with themselves");
  const int w = tk.w;
  // (MIXED: the geometry the row code sees changes with the phase; otherwise these are the constants they always were)
  const bool mixed = MIXED && (tk.qlen != tkb.qlen || tk.tlen != tkb.tlen);  // wave-uniform
  const int qmax = MIXED && tkb.qlen > tk.qlen ? tkb.qlen : tk.qlen, tmax = MIXED && tkb.tlen > tk.tlen ? tkb.tlen : tk.tlen;
  typename std::conditional<MIXED, int, const int>::type qlen = qmax, tlen = tmax;
  // Sequence windows in LDS.  Tb[i] = target position tt0 + i (bytes A | B << 8, zero beyond the ends); W[i] =
  // entry we0 + i of the reversed query with a 32-element front pad (entry j = QR[j-32], QR[e] = query[qlen-1-e];
  // A | B << 16, ready to use).  Short sequences fit whole (tt0 = we0 = 0 for ever); of long ones only the part
  // the band is moving through is resident (1024 entries of slack) and the windows are re-filled from the packed
  // pool when a block start finds them too far behind: the LDS footprint does not grow with the sequence length.
  const int tcap = pair_tcap(tmax, NREG), qcap = pair_qcap(qmax, NREG);
  uint16_t *Tb = reinterpret_cast<uint16_t *>(lds);
  uint32_t *W = reinterpret_cast<uint32_t *>(lds + 2 * tcap);
  // what the two halves of the windows are filled from (MIXED: per phase -- (A, B) with their own lengths, then (A, A), then (B, B))
  typename std::conditional<MIXED, int64_t, const int64_t>::type tw_a = tk.t_word, tw_b = tkb.t_word, qw_a = tk.q_word, qw_b = tkb.q_word;
  typename std::conditional<MIXED, int, const int>::type tl_a = tk.tlen, tl_b = MIXED ? tkb.tlen : tk.tlen, ql_a = tk.qlen,
                                                        ql_b = MIXED ? tkb.qlen : tk.qlen;
  int tt0_v = 0, we0_v = 0;  // window origins (always 0 without STREAM)
#define tt0 (STREAM ? tt0_v : 0)
#define we0 (STREAM ? we0_v : 0)
  auto fill_target = [&](const int from) {  // (pointers rebuilt here: re-fills are rare, registers are not)
    const uint32_t *twa = pool + tw_a, *tna = twa + (tl_a + 15) / 16;
    const uint32_t *twb = pool + tw_b, *tnb = twb + (tl_b + 15) / 16;
    tt0_v = from;
    for (int i = lane; i < tcap; i += 64) {
      const int t = from + i;
      if (MIXED) {  // (each half ends with its own target: zeroes beyond, as the reference's calloc'ed arena, :83-85)
        const uint32_t ca = t < tl_a ? pool_code8(twa, tna, t, sc.wild) : 0u, cb = t < tl_b ? pool_code8(twb, tnb, t, sc.wild) : 0u;
        Tb[i] = (uint16_t)(ca | (cb << 8));
      } else {
        Tb[i] = t < tlen ? (uint16_t)(pool_code8(twa, tna, t, sc.wild) | (pool_code8(twb, tnb, t, sc.wild) << 8)) : 0;
      }
    }
  };
  auto fill_query = [&](const int from) {
    const uint32_t *qwa = pool + qw_a, *qna = qwa + (ql_a + 15) / 16;
    const uint32_t *qwb = pool + qw_b, *qnb = qwb + (ql_b + 15) / 16;
    we0_v = from;
    for (int i = lane; i < qcap; i += 64) {
      const int e = from + i - 32;
      if (MIXED) {
        // entry e of the window is QR[e] of a query of `qlen` bases; a shorter query lies qlen - ql entries higher, so that
        // the address of row r's codes, qlen - 1 - r + t, reads QR_x[ql_x - 1 - r + t] of either task (zeroes around it)
        const int ea = e - (qlen - ql_a), eb = e - (qlen - ql_b);
        const uint32_t ca = ea >= 0 && ea < ql_a ? pool_code8(qwa, qna, ql_a - 1 - ea, sc.wild) : 0u;
        const uint32_t cb = eb >= 0 && eb < ql_b ? pool_code8(qwb, qnb, ql_b - 1 - eb, sc.wild) : 0u;
        W[i] = pair_qsel(ca, cb);
      } else {
        const bool in = e >= 0 && e < qlen;
        W[i] = in ? pair_qsel(pool_code8(qwa, qna, qlen - 1 - e, sc.wild), pool_code8(qwb, qnb, qlen - 1 - e, sc.wild)) : pair_qsel(0u, 0u);
      }
    }
  };

  // ---- unpack the 2-bit / N-mask sequences of both tasks into LDS ----
  int has_n;
  {
    const uint32_t *tna = pool + tw_a + (tl_a + 15) / 16, *tnb = pool + tw_b + (tl_b + 15) / 16;
    const uint32_t *qna = pool + qw_a + (ql_a + 15) / 16, *qnb = pool + qw_b + (ql_b + 15) / 16;
    uint32_t n_seen = 0;
    if (MIXED) {
      for (int k = lane; k < (tl_a + 31) / 32; k += 64) n_seen |= tna[k];
      for (int k = lane; k < (tl_b + 31) / 32; k += 64) n_seen |= tnb[k];
      for (int k = lane; k < (ql_a + 31) / 32; k += 64) n_seen |= qna[k];
      for (int k = lane; k < (ql_b + 31) / 32; k += 64) n_seen |= qnb[k];
    } else {
      for (int k = lane; k < (tlen + 31) / 32; k += 64) n_seen |= tna[k] | tnb[k];
      for (int k = lane; k < (qlen + 31) / 32; k += 64) n_seen |= qna[k] | qnb[k];
    }
    has_n = __builtin_amdgcn_readfirstlane((int)__any(n_seen != 0));  // wave-uniform
    fill_target(0);
    // row 0 reads entries up to qlen + NSLOT + 31: the window's top there
    fill_query(qlen + NSLOT + 36 > qcap ? qlen + NSLOT + 36 - qcap : 0);
  }
  __syncthreads();

  // ---- constants of the <<8 difference domain ----
  const unsigned qb2 = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned qv = qb2;
  const unsigned capv = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned zb_match = (unsigned)((sc.sc_match + sc.qe2_b) & 0xff), zb_mis = (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff);
  const unsigned t_mis4 = zb_mis * 0x01010101u, t_delta = zb_match ^ zb_mis, t_wild4 = (unsigned)sc.qe2_b * 0x01010101u;
  const unsigned z_wild = ((unsigned)sc.qe2_b << 8) * 0x00010001u;  // score 0, also "never written"
  unsigned one2 = 0x00010001u;  // min(x, 1) per half; opaque so that it stays one v_pk_min_u16 (in a scalar register)
  asm("" : "+s"(one2));

  unsigned U[NREG], V[NREG], X[NREG], Y[NREG], S[NREG], TA[NREG], TB[NREG];
  // the score tables of the lane's slot of register k from its entry of the target window (bytes A | B << 8; N: bit 7)
  auto load_target = [&](const int k, const unsigned tb16) {
    const unsigned ca = tb16 & 0xffu, cb = (tb16 >> 8) & 0xffu;
    TA[k] = (ca & 0x80u) ? t_wild4 : t_mis4 ^ (t_delta << (8u * ca));
    TB[k] = (cb & 0x80u) ? t_wild4 : t_mis4 ^ (t_delta << (8u * cb));
  };
  unsigned Fa[NREG], Fb[NREG], Fx[NREG], Fy[NREG];
  unsigned xt1[NREG], vt1[NREG];  // x, v of the (r-1, t-1) neighbours; persistent so that the two-step DPP shift
                                  // writes in place (every lane is overwritten each row)
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
    xt1[k] = vt1[k] = 0u;
    U[k] = V[k] = X[k] = Y[k] = 0u;
    S[k] = z_wild;
    Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    load_target(k, (unsigned)Tb[64 * k + lane]);
  }
  // TRACK: exact H of the lane's cell of task A per register, and the best (H, row) the lane has seen
  int32_t Hc[TRACK ? NREG : 1], bestH[TRACK ? NREG : 1], bestR[TRACK ? NREG : 1];
#pragma unroll
  for (int k = 0; k < (TRACK ? NREG : 1); ++k) {
    Hc[k] = SDF_NEG_INF;
    bestH[k] = 0;  // ksw_reset_extz: max = 0, so only H > 0 is ever recorded (extern/ksw2.h:155,:165)
    bestR[k] = -1;
  }

  const bool with_dir = !(tk.flag & SDF_FLAG_SCORE_ONLY);
  typename std::conditional<MIXED, uint2 *, uint2 *const>::type dir_a = reinterpret_cast<uint2 *>(dirbase + tk.dir_off),
                                                              dir_b = reinterpret_cast<uint2 *>(dirbase + tkb.dir_off);
  typename std::conditional<MIXED, int, const int>::type nrow = qlen + tlen - 1, out_a = tk.out_idx, out_b = tkb.out_idx;
#define bperm_idx (((lane + 16) & 63) * 4)  // (used once per re-base: not worth a register)

  int base = 0;
  int prev_lo = -1;
  unsigned carry_x = 0u, carry_v = 0u;  // packed (r-1) values shifted into slot 0 on the first row of a block
  bool zero_low = false;  // slots below the reference window still hold x,v that must read as 0
  int32_t h_top[2] = {0, 0}, h_under[2] = {0, 0};  // H of the top cell / of the cell the next top cell reads
  int32_t ez_score[2] = {SDF_NEG_INF, SDF_NEG_INF}, ez_mte[2] = {SDF_NEG_INF, SDF_NEG_INF}, ez_mte_q[2] = {-1, -1};
  int32_t ez_zdropped = 0;
  int drop_row = -1;  // row of the current block at which the reference window left slots 0..15
  int r0 = 0;
  unsigned qaddr = 0u;  // LDS address of the query codes of the lean row about to be computed
  unsigned hacc_a = 0u, hacc_b = 0u;  // lane-distributed parts of the H path sums (lean rows), folded lazily
  int hcnt = 0;                       // number of path steps in them (each subtracts q+e)
  auto fold_h = [&]() {  // bring the scalar path values up to date
    if (hcnt) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        hacc_a += (unsigned)__shfl_xor((int)hacc_a, off);
        hacc_b += (unsigned)__shfl_xor((int)hacc_b, off);
      }
      h_under[0] += (int32_t)hacc_a - hcnt * sc.qe;
      h_under[1] += (int32_t)hacc_b - hcnt * sc.qe;
      h_top[0] = h_under[0];
      h_top[1] = h_under[1];
      hacc_a = hacc_b = 0u;
      hcnt = 0;
    }
  };
  // scalar H bookkeeping of one row for both tasks from the packed u (top cell) / v (cell under it)
  auto h_row = [&](const int r, const int hi0, const int lo0, const int hi, const bool first, const bool want_top,
                   const bool top_from_under, const bool up, const unsigned uh, const unsigned vu) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int32_t uht = (int32_t)((uh >> (16 * t + 8)) & 0xffu), vut = (int32_t)((vu >> (16 * t + 8)) & 0xffu);
      if (want_top) {
        if (first) h_top[t] = uht - 2 * sc.qe;
        else h_top[t] = (top_from_under ? h_under[t] : h_top[t]) + uht - sc.qe;
      }
      if (up || first) {
        h_under[t] = h_top[t];
      } else if (hi0 - 1 >= lo0) {
        h_under[t] += vut - sc.qe;
      }
      if (hi0 == tlen - 1) {
        if (h_top[t] > ez_mte[t]) {
          ez_mte[t] = h_top[t];
          ez_mte_q[t] = r - hi;
        }
        if (r == nrow - 1) ez_score[t] = h_top[t];
      }
    }
  };

  int32_t carry_h = SDF_NEG_INF;  // TRACK: H of target position base - 1 (left the window at the last re-base)
  // TRACK: H of every in-band cell after row r's recurrence (reference :222-258), and the running best per lane
  auto track_row = [&](const int r, const int lo0, const int hi0) {
    if (!TRACK) return;
    const int st = hi0 - base;  // slot of the top cell
    // H of the cell under it BEFORE this row's update; under slot 0 lies the cell the last re-base pushed out
    int32_t h_below = st == 0 ? carry_h : 0;
#pragma unroll
    for (int k = 0; k < (TRACK ? NREG : 0); ++k)
      if (st > 0 && ((st - 1) >> 6) == k) h_below = __builtin_amdgcn_readlane(Hc[k], (st - 1) & 63);
#pragma unroll
    for (int k = 0; k < (TRACK ? NREG : 0); ++k) {
      const int t = base + 64 * k + lane;
      if (base + 64 * k > hi0 || base + 64 * k + 63 < lo0) continue;  // (wave-uniform) no in-band cell in this register
      const int32_t vA = (int32_t)((V[k] >> 8) & 0xffu), uA = (int32_t)((U[k] >> 8) & 0xffu);
      const bool mid = (unsigned)(t - lo0) < (unsigned)(hi0 - lo0);  // lo0 <= t < hi0
      const bool top = t == hi0;
      int32_t h = Hc[k];
      h = mid ? h + vA - sc.qe : h;
      const int32_t htop = r == 0 ? vA - 2 * sc.qe : (hi0 > 0 ? h_below + uA : h + vA) - sc.qe;
      h = top ? htop : h;
      Hc[k] = h;
      const bool better = (mid || top) && h > bestH[k];
      bestH[k] = better ? h : bestH[k];
      bestR[k] = better ? r : bestR[k];
    }
  };

  // ------------------------------------------------------------------------------------------
  // General row: every special case of the reference (first/last rows, boundary cell t = r,
  // clipping by the sequence ends, carry-in artefacts).  Returns false when the band is exhausted.
  // ------------------------------------------------------------------------------------------
  auto slow_row = [&](const int r) -> bool {
    fold_h();
    int lo0 = (r - w + 1) >> 1, hi0 = (r + w) >> 1;
    lo0 = lo0 < r - qlen + 1 ? r - qlen + 1 : lo0;
    lo0 = lo0 < 0 ? 0 : lo0;
    hi0 = hi0 > r ? r : hi0;
    hi0 = hi0 > tlen - 1 ? tlen - 1 : hi0;
    if (lo0 > hi0) return false;
    const int lo = lo0 & ~15, hi = hi0 | 15;
    const int off_lo = lo - base;  // 0 or 16
    const int off_hi = hi - base;  // last enabled slot
    // the reference rebased at this row: slot off_lo's (r-1,t-1) neighbour is slot 15 (natural);
    // on later rows that neighbour reads as 0
    const bool ref_rebased = lo != prev_lo && prev_lo >= 0;
    if (ref_rebased && off_lo == 16) drop_row = r;
    if (off_lo == 16 && !ref_rebased && !zero_low) {
      if (lane < 16) {
        X[0] = 0u;
        V[0] = 0u;
      }
      zero_low = true;
    }
    // ---- boundary cell t = r: y = 0, u = gap open (reference :122) ----
    if (hi >= r) {
      const int sr = r - base;
      const unsigned uval = r ? qb2 : 0u;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {  // selects on every register: conditional stores into the arrays would be
        const bool mine = (sr >> 6) == k && lane == (sr & 63);  // merged into one dynamically indexed store
        U[k] = mine ? uval : U[k];
        Y[k] = mine ? 0u : Y[k];
      }
    }
    // ---- (r-1, t-1) neighbours: shift x and v up by one slot ----
    {
      // carry into slot 0: x = 0, v = gap open when the window starts at t = 0 (r > 0); the
      // captured (r-1) values when the reference re-bases exactly at a block start
      const unsigned vcarry = (base == 0 && r > 0) ? qb2 : (r == r0 ? carry_v : 0u);
      const unsigned xcarry = (base != 0 && r == r0) ? carry_x : 0u;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (k == 0) {
          xt1[0] = (unsigned)__builtin_amdgcn_update_dpp((int)xcarry, (int)X[0], 0x138, 0xf, 0xf, false);
          vt1[0] = (unsigned)__builtin_amdgcn_update_dpp((int)vcarry, (int)V[0], 0x138, 0xf, 0xf, false);
        } else {
          const int x0 = __builtin_amdgcn_mov_dpp((int)X[k - 1], 0x13C, 0x1, 0x1, false);
          xt1[k] = (unsigned)__builtin_amdgcn_update_dpp(x0, (int)X[k], 0x138, 0xf, 0xf, false);
          const int v0 = __builtin_amdgcn_mov_dpp((int)V[k - 1], 0x13C, 0x1, 0x1, false);
          vt1[k] = (unsigned)__builtin_amdgcn_update_dpp(v0, (int)V[k], 0x138, 0xf, 0xf, false);
        }
      }
      // sign-extension artefact of the reference's carry-in (:145-146): a negative v carry also
      // sets lanes 1..3 of the first block.  Only possible on the reference's rebase rows.
      if (ref_rebased) {
        if (off_lo == 16) {
          const unsigned sm = sign_smear((unsigned)__builtin_amdgcn_readlane((int)V[0], 15));
          if (sm && lane_in(lane, 17, 20)) vt1[0] |= sm;
        } else if (r == r0) {
          const unsigned sm = sign_smear(carry_v);
          if (sm && lane_in(lane, 1, 4)) vt1[0] |= sm;
        }
      }
    }
    // ---- scores: refresh [lo0, lo0 + 16*n), keep the old value elsewhere ----
    {
      const int ra = lo0 - base;
      const int rb = ra + ((hi0 - lo0) & ~15) + 16;
      const int cq = qlen - 1 - r + base + 32;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        const int a_ = ra - 64 * k, b_ = rb - 64 * k;
        if (b_ > 0 && a_ < 64) {
          const unsigned qc = W[cq - we0 + 64 * k + lane];
          unsigned z;
          SDF_PFRESH(z, k, qc, has_n)
          if (a_ <= 0 && b_ >= 64) S[k] = z;
          else if (lane_in(lane, a_ < 0 ? 0 : a_, b_)) S[k] = z;
        }
      }
    }
    // ---- the recurrence on the reference's widened range [lo, hi] ----
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int l0 = off_lo - 64 * k <= 0 ? 0 : off_lo - 64 * k;
      const int l1 = off_hi - 64 * k;
      if (l1 >= l0 && l0 < 64) {
        if ((unsigned)(lane - l0) <= (unsigned)(l1 - l0)) SDF_CORE(k)
      }
    }
    // ---- exact H of the top cell and of the cell under the band edge (score, mte) ----
    {
      const int st = hi0 - base;  // slot of the top cell
      unsigned uh = 0, vu = 0;
      int hin = (r + 1 + w) >> 1;  // next row's top cell: does it move up?
      hin = hin > r + 1 ? r + 1 : hin;
      hin = hin > tlen - 1 ? tlen - 1 : hin;
      const bool up = hin == hi0 + 1 || hin == 0;
      const bool want_top = up || hi0 == tlen - 1 || hi0 == 0;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (want_top && (st >> 6) == k) {  // two scalar reads, scalar select (no select between register arrays)
          const unsigned ru = (unsigned)__builtin_amdgcn_readlane((int)U[k], st & 63);
          const unsigned rv = (unsigned)__builtin_amdgcn_readlane((int)V[k], st & 63);
          uh = hi0 > 0 ? ru : rv;
        }
        if (!up && st > 0 && ((st - 1) >> 6) == k) vu = (unsigned)__builtin_amdgcn_readlane((int)V[k], (st - 1) & 63);
      }
      h_row(r, hi0, lo0, hi, r == 0, want_top, hi0 > 0, up, uh, vu);
    }
    track_row(r, lo0, hi0);
    prev_lo = lo;
    return true;
  };

  // ------------------------------------------------------------------------------------------
  // Lean rows [rb, re) of one block (rb >= 1): the general row with the rare cases taken out and
  // everything constant over the segment hoisted (see extz2_wave.hip for the LOW16 / SCALARH /
  // STEADY regimes).  Lanes above the window top are NOT masked; the caller zeroes them when the
  // window grows over them.
  // ------------------------------------------------------------------------------------------
  auto lean_rows_n = [&](auto low16_c, auto scalarh_c, auto steady_c, auto hasn_c, const int rb, const int re) {
    constexpr bool HASN = decltype(hasn_c)::value;  // N handling compiled in or out (no per-row test)
    constexpr bool LOW16 = decltype(low16_c)::value;
    constexpr bool SCALARH = decltype(scalarh_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    constexpr int KT = NREG - 1;
    if (SCALARH) fold_h();
    // (the row's query codes are read at its start: a fetch one row ahead costs four register copies a row and three
    // registers, and with four or five wavefronts on the SIMD the LDS latency is covered anyway)
    qaddr = (unsigned)(2 * tcap + 4 * (qlen - 1 - rb + base + 32 - we0 + lane));
    if (STEADY && !SCALARH) hcnt += re - rb;  // every steady row takes one path step
    const unsigned vcar = base == 0 ? qb2 : 0u;  // v carry into slot 0 (r > 0)
#pragma unroll 1
    for (int r = rb; r < re; ++r) {
      int hi0 = (r + w) >> 1, lo0 = (r - w + 1) >> 1;
      if (!STEADY) {
        lo0 = lo0 < r - qlen + 1 ? r - qlen + 1 : lo0;
        lo0 = lo0 < 0 ? 0 : lo0;
        hi0 = hi0 > r ? r : hi0;
        hi0 = hi0 > tlen - 1 ? tlen - 1 : hi0;
      }
      const int off_hi = (hi0 | 15) - base;
      unsigned qcur[NREG];
#pragma unroll
      for (int k = 0; k < NREG; ++k) qcur[k] = *reinterpret_cast<const uint32_t *>(lds + qaddr + 256 * k);
      qaddr -= 4;
      // boundary cell t = r: y = 0, u = gap open (reference :122)
      if (!STEADY && off_hi + base >= r) {
        const int sr = r - base;
#pragma unroll
        for (int k = 0; k < NREG; ++k) {  // selects on every register (see slow_row)
          const bool mine = (sr >> 6) == k && lane == (sr & 63);
          U[k] = mine ? qb2 : U[k];
          Y[k] = mine ? 0u : Y[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (k == 0) {
          xt1[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X[0], 0x138, 0xf, 0xf, true);
          if (STEADY) vt1[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)V[0], 0x138, 0xf, 0xf, true);
          else vt1[0] = (unsigned)__builtin_amdgcn_update_dpp((int)vcar, (int)V[0], 0x138, 0xf, 0xf, false);
        } else {  // lane 0 from the register below, lanes 1..63 from this one: all lanes overwritten in place
          const int x0 = __builtin_amdgcn_mov_dpp((int)X[k - 1], 0x13C, 0x1, 0x1, false);
          xt1[k] = (unsigned)__builtin_amdgcn_update_dpp(x0, (int)X[k], 0x138, 0xf, 0xf, false);
          const int v0 = __builtin_amdgcn_mov_dpp((int)V[k - 1], 0x13C, 0x1, 0x1, false);
          vt1[k] = (unsigned)__builtin_amdgcn_update_dpp(v0, (int)V[k], 0x138, 0xf, 0xf, false);
        }
      }
      // scores: refreshed slots are [ra, rbe)
      const int ra = lo0 - base;
      const int rbe = ra + ((hi0 - lo0) & ~15) + 16;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        const int b_ = rbe - 64 * k;
        if (STEADY) {
          unsigned z;
          SDF_PFRESH(z, k, qcur[k], HASN)
          if (NREG == 1) S[0] = in_mask(lane_mask(ra, b_)) ? z : S[0];
          else if (k == 0) S[0] = in_mask(~0ull << ra) ? z : S[0];  // (steady: 0 <= ra < 32 -- one scalar shift, not lane_mask's six)
          else if (k == KT) S[k] = in_mask(b_ <= 0 ? 0ull : ~0ull >> (64 - (b_ < 64 ? b_ : 64))) ? z : S[k];
          else S[k] = z;
        } else if (b_ > 0) {
          unsigned z;
          SDF_PFRESH(z, k, qcur[k], HASN)
          if (k == 0) {
            if (b_ >= 64) S[0] = lane >= ra ? z : S[0];
            else S[0] = lane_in(lane, ra, b_) ? z : S[0];
          } else if (b_ >= 64) {
            S[k] = z;
          } else {
            S[k] = lane < b_ ? z : S[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (STEADY || off_hi >= 64 * k) {
          if (k == 0 && LOW16) {
            if (lane >= 16) SDF_CORE(0)
          } else {
            SDF_CORE(k)
          }
        }
      }
      if (SCALARH) {
        // top cell H every row: h_top = H(cell under the edge, previous row) + u(top) - (q+e)
        const int st = hi0 - base;
        unsigned uh = 0u, vu = 0u;
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          if ((st >> 6) == k) uh = (unsigned)__builtin_amdgcn_readlane((int)U[k], st & 63);
          if (((st - 1) >> 6) == k) vu = (unsigned)__builtin_amdgcn_readlane((int)V[k], (st - 1) & 63);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          h_top[t] = h_under[t] + (int32_t)((uh >> (16 * t + 8)) & 0xffu) - sc.qe;
          if (hi0 - 1 >= lo0) h_under[t] += (int32_t)((vu >> (16 * t + 8)) & 0xffu) - sc.qe;
          if (h_top[t] > ez_mte[t]) {
            ez_mte[t] = h_top[t];
            ez_mte_q[t] = r - (hi0 | 15);
          }
          if (r == nrow - 1) ez_score[t] = h_top[t];
        }
      } else {
        // H path: rows whose successor moves the top cell up read u of the top cell, the others
        // read v of the cell under it.  Added up inside the owning lane, reduced once at the end.
        int up;
        if (STEADY) {
          up = (r + w) & 1;
        } else {
          int hin = (r + 1 + w) >> 1;
          hin = hin > r + 1 ? r + 1 : hin;
          hin = hin > tlen - 1 ? tlen - 1 : hin;
          up = hin == hi0 + 1;
        }
        if (STEADY || up || hi0 - 1 >= lo0) {
          const int sl = hi0 - base - 1 + up;
          // selects between VALUES read first: a select between the arrays themselves would turn them
          // into an indexed stack object
          unsigned val = 0u;
          if (STEADY) {
            constexpr int KL = KT > 0 ? KT - 1 : 0;
            const unsigned u_hi = U[KT], v_hi = V[KT], u_lo = U[KL], v_lo = V[KL];
            const unsigned c_hi = up ? u_hi : v_hi, c_lo = up ? u_lo : v_lo;
            val = (NREG > 1 && sl < 64 * KT) ? c_lo : c_hi;
          } else {
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
              const unsigned uk = U[k], vk = V[k];
              const unsigned ck = up ? uk : vk;
              val = (sl >> 6) == k ? ck : val;
            }
          }
          if (in_mask(1ull << (sl & 63))) {
            hacc_a += (val >> 8) & 0xffu;
            hacc_b += val >> 24;
          }
          if (!STEADY) ++hcnt;
        }
      }
      track_row(r, lo0, hi0);
    }
  };
  auto lean_rows = [&](auto low16_c, auto scalarh_c, auto steady_c, const int rb, const int re) {
    if (has_n) lean_rows_n(low16_c, scalarh_c, steady_c, std::true_type{}, rb, re);
    else lean_rows_n(low16_c, scalarh_c, steady_c, std::false_type{}, rb, re);
  };
  // U,V,X,Y of the cells t in [t_from, t_to] back to "never computed" (both bounds block aligned)
  auto zero_cells = [&](const int t_from, const int t_to) {
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int a_ = t_from - base - 64 * k, b_ = t_to - base - 64 * k;
      if (b_ >= 0 && a_ < 64) {
        const int la = a_ <= 0 ? 0 : a_;
        if ((unsigned)(lane - la) <= (unsigned)(b_ - la)) {
          U[k] = 0u;
          V[k] = 0u;
          X[k] = 0u;
          Y[k] = 0u;
        }
      }
    }
  };
  int win_hi = -1;    // last cell of the reference window so far (cells above it were never computed)
  int dirty_hi = -1;  // cells in (win_hi, dirty_hi] may hold scratch values left by lean rows

  // MIXED, tasks of different lengths: rows [0, shared) together (phase 0), then [shared, nrow) of task A (phase 1) and of
  // task B (phase 2), each with the task in both halves.  State of half B across phase 1: in LDS behind the windows.
  const int shared = mixed ? pair_shared_rows(tk.qlen, tk.tlen, tkb.qlen, tkb.tlen, w) : 0;
  uint32_t *keep = reinterpret_cast<uint32_t *>(lds + ((2 * tcap + 4 * qcap + 15) & ~15));  // 5 x NREG x 64 words (MIXED launches only)
  int kp_base = 0, kp_prev_lo = 0, kp_win_hi = 0, kp_dirty_hi = 0, kp_h_top = 0, kp_h_under = 0, kp_score = 0, kp_mte = 0, kp_mte_q = 0;
  bool kp_zero_low = false;
#pragma nounroll
  for (int phase = 0; phase < (mixed ? 3 : 1); ++phase) {
  int row_first = 0, row_last = nrow;
  if constexpr (MIXED) {
    if (!mixed) {
    } else if (phase == 0) {
      row_last = shared;
    } else {
      fold_h();
      if (phase == 1) {  // half B waits; half A goes on in both halves
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          keep[(0 * NREG + k) * 64 + lane] = U[k];
          keep[(1 * NREG + k) * 64 + lane] = V[k];
          keep[(2 * NREG + k) * 64 + lane] = X[k];
          keep[(3 * NREG + k) * 64 + lane] = Y[k];
          keep[(4 * NREG + k) * 64 + lane] = S[k];
          U[k] = __builtin_amdgcn_perm(U[k], U[k], 0x01000100u);
          V[k] = __builtin_amdgcn_perm(V[k], V[k], 0x01000100u);
          X[k] = __builtin_amdgcn_perm(X[k], X[k], 0x01000100u);
          Y[k] = __builtin_amdgcn_perm(Y[k], Y[k], 0x01000100u);
          S[k] = __builtin_amdgcn_perm(S[k], S[k], 0x01000100u);
        }
        kp_base = base, kp_prev_lo = prev_lo, kp_win_hi = win_hi, kp_dirty_hi = dirty_hi, kp_zero_low = zero_low;
        kp_h_top = h_top[1], kp_h_under = h_under[1], kp_score = ez_score[1], kp_mte = ez_mte[1], kp_mte_q = ez_mte_q[1];
        h_top[1] = h_top[0], h_under[1] = h_under[0], ez_score[1] = ez_score[0], ez_mte[1] = ez_mte[0], ez_mte_q[1] = ez_mte_q[0];
        tw_b = tw_a, qw_b = qw_a, tl_b = tl_a, ql_b = ql_a;
        dir_b = dir_a;
        out_b = out_a;
      } else {  // task B, as phase 0 left it, in both halves
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          const unsigned u = keep[(0 * NREG + k) * 64 + lane], v = keep[(1 * NREG + k) * 64 + lane];
          const unsigned x = keep[(2 * NREG + k) * 64 + lane], y = keep[(3 * NREG + k) * 64 + lane];
          const unsigned sv = keep[(4 * NREG + k) * 64 + lane];
          U[k] = __builtin_amdgcn_perm(u, u, 0x03020302u);
          V[k] = __builtin_amdgcn_perm(v, v, 0x03020302u);
          X[k] = __builtin_amdgcn_perm(x, x, 0x03020302u);
          Y[k] = __builtin_amdgcn_perm(y, y, 0x03020302u);
          S[k] = __builtin_amdgcn_perm(sv, sv, 0x03020302u);
        }
        base = kp_base, prev_lo = kp_prev_lo, win_hi = kp_win_hi, dirty_hi = kp_dirty_hi, zero_low = kp_zero_low;
        h_top[0] = h_top[1] = kp_h_top, h_under[0] = h_under[1] = kp_h_under;
        ez_score[0] = ez_score[1] = kp_score, ez_mte[0] = ez_mte[1] = kp_mte, ez_mte_q[0] = ez_mte_q[1] = kp_mte_q;
        tw_a = tw_b = tkb.t_word, qw_a = qw_b = tkb.q_word, tl_a = tl_b = tkb.tlen, ql_a = ql_b = tkb.qlen;
        dir_a = dir_b = reinterpret_cast<uint2 *>(dirbase + tkb.dir_off);
        out_a = out_b = tkb.out_idx;
      }
      qlen = ql_a, tlen = tl_a;
      nrow = qlen + tlen - 1;
      row_first = shared, row_last = nrow;
      // the windows again, from this task alone in its own layout (the block start below re-bases and checks the query window)
      __syncthreads();
      fill_target(STREAM ? base : 0);
      if (STREAM) we0_v = 1 << 28;  // (out of reach: the block start fills the query window where this row needs it)
      else fill_query(0);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NREG; ++k) load_target(k, (unsigned)Tb[base - tt0 + 64 * k + lane]);
    }
  }
  for (r0 = row_first; r0 < row_last && !ez_zdropped; r0 += 16) {
    // ---- block start: re-base the window to the reference's band start of this row ----
    {
      Band b0;
      if (!band_of(r0, qlen, tlen, w, b0)) {
        ez_zdropped = 1;
        break;
      }
      carry_x = carry_v = 0u;
      if (b0.lo != base) {  // always +16: shift everything down by 16 lanes
        if (prev_lo == base) {  // the reference re-bases at this very row: its carry-in is the
          carry_x = (unsigned)__builtin_amdgcn_readlane((int)X[0], 15);  // (r-1) value of the cell just
          carry_v = (unsigned)__builtin_amdgcn_readlane((int)V[0], 15);  // below the new window
        }
        if (TRACK) carry_h = __builtin_amdgcn_readlane(Hc[0], 15);
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          const bool from_next = lane >= 48;
          unsigned a0, a1;
#define SDF_SHIFT16(A, INIT)                                                             \
  a0 = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_idx, (int)A[k]);                     \
  a1 = (k + 1 < NREG) ? (unsigned)__builtin_amdgcn_ds_bpermute(bperm_idx, (int)A[k + 1 < NREG ? k + 1 : k]) : (INIT); \
  A[k] = from_next ? a1 : a0;
          SDF_SHIFT16(U, 0u)
          SDF_SHIFT16(V, 0u)
          SDF_SHIFT16(X, 0u)
          SDF_SHIFT16(Y, 0u)
          SDF_SHIFT16(S, z_wild)
          if (TRACK) {
            const int h0 = __builtin_amdgcn_ds_bpermute(bperm_idx, Hc[TRACK ? k : 0]);
            const int h1 = (k + 1 < NREG) ? __builtin_amdgcn_ds_bpermute(bperm_idx, Hc[TRACK && k + 1 < NREG ? k + 1 : 0]) : SDF_NEG_INF;
            Hc[TRACK ? k : 0] = from_next ? h1 : h0;
          }
#undef SDF_SHIFT16
        }
        base = b0.lo;
        if (STREAM && __builtin_expect(base + NSLOT > tt0 + tcap, 0)) {  // the band has moved beyond the resident part of the target
          fill_target(base);
          __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < NREG; ++k)
          load_target(k, (unsigned)Tb[base - tt0 + 64 * k + lane]);
        zero_low = false;
      }
      {  // reversed-query entries this block reads (rows r0 .. r0+16, the last one as a prefetch): resident?
        const int e_lo = qlen - 1 - (r0 + 16) + base + 32, e_hi = qlen - 1 - r0 + base + 32 + NSLOT - 1;
        if (STREAM && __builtin_expect(e_lo < we0 || e_hi >= we0 + qcap, 0)) {
          // they move towards lower entries as the rows advance: the block's range goes to the window's top
          const int from = e_hi + 1 - qcap;
          fill_query(from < 0 ? 0 : from);
          __syncthreads();
        }
      }
    }
    const int rend = r0 + 16 < nrow ? r0 + 16 : nrow;
    drop_row = -1;
    int r = r0;
    // rows of this block: lean segments between the rows at which the reference window changes
    {
      constexpr int KT = NREG - 1;
      const int rl = r0 + 15;
      // pure band regime on all 16 rows, no boundary cell, refresh range spanning registers 0..KT
      bool steady = w >= 2 && r0 + 16 <= nrow && base >= 16 && ((rl - w + 1) >> 1) >= rl - qlen + 1 &&
                    ((rl + w) >> 1) < tlen - 1 && ((r0 + w) >> 1) + 15 < r0;
      if (steady) {
        const int lo0a = (r0 - w + 1) >> 1, hi0a = (r0 + w) >> 1;
        steady = lo0a + ((w - 1) & ~15) + 16 - base >= 64 * KT && (hi0a | 15) - base >= 64 * KT &&
                 hi0a - 1 - base >= 64 * KT - 64;
      }
      const bool lean_ok = tlen >= 2 && w >= 1;
      bool low16 = false;
      while (r < rend) {
        int lo0 = (r - w + 1) >> 1, hi0 = (r + w) >> 1;
        lo0 = lo0 < r - qlen + 1 ? r - qlen + 1 : lo0;
        lo0 = lo0 < 0 ? 0 : lo0;
        hi0 = hi0 > r ? r : hi0;
        hi0 = hi0 > tlen - 1 ? tlen - 1 : hi0;
        if (lo0 > hi0) {
          ez_zdropped = 1;
          break;
        }
        const int lo = lo0 & ~15, hi = hi0 | 15;
        if (hi > win_hi) {  // the window grows over cells that must read as "never computed"
          if (dirty_hi > win_hi) zero_cells(win_hi + 1, hi < dirty_hi ? hi : dirty_hi);
          win_hi = hi;
          if (dirty_hi < win_hi) dirty_hi = win_hi;
        }
        const bool rebase_row = lo != prev_lo && prev_lo >= 0;
        bool special = !lean_ok || r == 0 || (r == r0 && (carry_x | carry_v) != 0u);
        if (rebase_row && !special) {
          // natural neighbour, but mind the sign-extension artefact of a negative carry (either task)
          const unsigned cvh = lo - base == 16 ? (unsigned)__builtin_amdgcn_readlane((int)V[0], 15) : carry_v;
          special = (cvh & 0x80008000u) != 0u;
        }
        if (special) {
          if (dirty_hi > win_hi) zero_cells(win_hi + 1, dirty_hi);
          dirty_hi = win_hi;
          if (!slow_row(r)) {
            ez_zdropped = 1;
            break;
          }
          low16 = prev_lo - base == 16;
          ++r;
          continue;
        }
        if (rebase_row) {
          if (lo - base == 16) {
            drop_row = r;
            low16 = true;
          }
        } else if (low16 && !zero_low) {
          if (lane < 16) {
            X[0] = 0u;
            V[0] = 0u;
          }
          zero_low = true;
        }
        // rows until the reference window changes again (closed forms of the band geometry)
        int stop = rend;
        if (rebase_row) {
          stop = r + 1;  // the re-base row runs alone: slots 0..15 are zeroed right after it
        } else {
          int rr = lo + 15 + qlen;
          const int rr2 = 2 * (lo + 16) + w - 1;
          rr = rr2 < rr ? rr2 : rr;
          if (rr > r && rr < stop) stop = rr;
          const int h1 = hi + 1;
          if (h1 <= tlen - 1) {
            int rh = 2 * h1 - w;
            rh = rh < h1 ? h1 : rh;
            if (rh > r && rh < stop) stop = rh;
          }
          int rt = 2 * (tlen - 1) - w;
          rt = rt < tlen - 1 ? tlen - 1 : rt;
          if (rt > r && rt < stop) stop = rt;
          if (TRACK) {  // the row at which the band runs out (lo0 > hi0) ends a segment too: the check above sees it
            const int rx = 2 * (tlen < qlen ? tlen : qlen) + w - 1;
            if (rx > r && rx < stop) stop = rx;
          }
        }
        const bool scalarh = hi0 == tlen - 1;
        if (scalarh) {
          if (low16) lean_rows(std::true_type{}, std::true_type{}, std::false_type{}, r, stop);
          else lean_rows(std::false_type{}, std::true_type{}, std::false_type{}, r, stop);
        } else if (steady) {
          if (low16) lean_rows(std::true_type{}, std::false_type{}, std::true_type{}, r, stop);
          else lean_rows(std::false_type{}, std::false_type{}, std::true_type{}, r, stop);
        } else {
          if (low16) lean_rows(std::true_type{}, std::false_type{}, std::false_type{}, r, stop);
          else lean_rows(std::false_type{}, std::false_type{}, std::false_type{}, r, stop);
        }
        {  // the top register of the window now holds scratch values above the window
          const int top = base + 64 * (((win_hi - base) >> 6) + 1) - 1;
          if (top > dirty_hi) dirty_hi = top;
        }
        prev_lo = lo;
        r = stop;
      }
      if (dirty_hi > win_hi) zero_cells(win_hi + 1, dirty_hi);  // clean lanes for the re-base shift
      dirty_hi = win_hi;
    }
    // ---- block end: direction flags of these (<=16) rows leave for HBM ----
    if (with_dir) {
      const int done = r - r0;
      const int rbk = r0 >> 4;
      if (drop_row >= 0 && lane < 16) {  // lanes that stopped shifting when their slots were dropped
        const unsigned sh = (unsigned)(r - drop_row);
        Fa[0] = pk_shl(Fa[0], sh);
        Fb[0] = pk_shl(Fb[0], sh);
        Fx[0] = pk_shl(Fx[0], sh);
        Fy[0] = pk_shl(Fy[0], sh);
      }
      if (done > 0) {
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          unsigned fa = Fa[k], fb = Fb[k], fx = Fx[k], fy = Fy[k];
          if (done < 16) {
            const unsigned sh = 16 - done;
            fa = pk_shl(fa, sh);
            fb = pk_shl(fb, sh);
            fx = pk_shl(fx, sh);
            fy = pk_shl(fy, sh);
          }
          // per task: (a | b << 16, x | y << 16), bit 15 - (r % 16) of each 16-bit mask is row r
          const int64_t at = ((int64_t)rbk * NREG + k) * 64 + lane;
          dir_a[at] = make_uint2(__builtin_amdgcn_perm(fb, fa, 0x05040100u), __builtin_amdgcn_perm(fy, fx, 0x05040100u));
          dir_b[at] = make_uint2(__builtin_amdgcn_perm(fb, fa, 0x07060302u), __builtin_amdgcn_perm(fy, fx, 0x07060302u));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
  }
  if (MIXED && mixed && phase == 0) continue;  // (results: at the end of each task's own phase)

  fold_h();
  int32_t ez_max = 0, ez_max_t = -1, ez_max_q = -1;
  if (TRACK) {
    BestCell best = {0, -1, 0, -1};
#pragma unroll
    for (int k = 0; k < (TRACK ? NREG : 0); ++k) {
      const int rr = bestR[k];
      if (rr < 0) continue;
      Band bb, br;
      band_of((rr >> 4) << 4, qlen, tlen, w, bb);  // window base of the row's block
      band_of(rr, qlen, tlen, w, br);
      const int t = bb.lo + 64 * k + lane;
      const int vec_end = br.lo0 + (br.hi0 - br.lo0) / 4 * 4;  // the reference's scan order inside a row (:226-258)
      const int key = t == br.hi0 ? 0 : t < vec_end ? 1 + (((t - br.lo0) & 3) << 20) + t : 1 + (4 << 20) + t;
      const BestCell cand = {bestH[k], rr, key, t};
      if (beats(cand, best)) best = cand;
    }
    best = wave_best(best);
    if (best.r >= 0) {
      ez_max = best.H;
      ez_max_t = best.t;
      ez_max_q = best.r - best.t;
    }
  }
  if (lane < 2) {
    const bool second = lane == 1;
    sdf_result o;
    o.score = second ? ez_score[1] : ez_score[0];
    o.max = ez_max;
    o.max_q = ez_max_q;
    o.max_t = ez_max_t;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = second ? ez_mte[1] : ez_mte[0];
    o.mte_q = second ? ez_mte_q[1] : ez_mte_q[0];
    o.zdropped = ez_zdropped;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[second ? out_b : out_a] = o;
  }
  }  // phase
}

#undef tt0
#undef we0
#undef bperm_idx

template <int NREG, bool STREAM, bool TRACK>
__global__ __launch_bounds__(64, NREG <= 1 ? 6 : NREG <= 2 ? 5 : NREG <= 3 ? 4 : NREG <= 4 ? 3 : 2) void extz2_pair_kernel(
    const PlanTask *__restrict__ plan, const int32_t *__restrict__ order, const uint32_t *__restrict__ pool,
    ScoreK sc, uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  pair_body<NREG, STREAM, TRACK>(plan, order, pool, sc, dirbase, res);
}
// The instantiation of the headline batch (w = 128: three registers, sequences whole in LDS) runs FIVE wavefronts per SIMD
// (96 VGPRs; LDS allows six): every row loop fits without a scratch access, eight registers of block-level state are
// spilled (32 bytes; one store and a few loads per 16-row block).  19.4-19.6 ms for the whole batch in one launch against
// 20.1-20.2 at four wavefronts (profiles/r03_shapes.txt).
template <>
__global__ __launch_bounds__(64, 5) void extz2_pair_kernel<3, false, false>(
    const PlanTask *__restrict__ plan, const int32_t *__restrict__ order, const uint32_t *__restrict__ pool,
    ScoreK sc, uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  pair_body<3, false, false>(plan, order, pool, sc, dirbase, res);
}

#define SDF_PAIR_INST(N)                                                                                               \
  template __global__ void extz2_pair_kernel<N, false, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, \
                                                              uint8_t *, sdf_result *);                               \
  template __global__ void extz2_pair_kernel<N, true, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,  \
                                                             uint8_t *, sdf_result *);
SDF_PAIR_INST(1)
SDF_PAIR_INST(2)
template __global__ void extz2_pair_kernel<3, true, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                           uint8_t *, sdf_result *);  // (<3, false, false>: specialised above)
SDF_PAIR_INST(4)
SDF_PAIR_INST(6)
SDF_PAIR_INST(8)
#undef SDF_PAIR_INST
// TRACK flavour (band-exhausting tasks; windows streamed): windows of up to 192 / 384 slots
template __global__ void extz2_pair_kernel<3, true, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                          uint8_t *, sdf_result *);
template __global__ void extz2_pair_kernel<6, true, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                          uint8_t *, sdf_result *);

// Mixed pairs (two tasks of one band and flag set, any lengths): windows always streamed (a window that holds its
// sequences whole never re-fills), 2 .. 9 registers of 64 slots -- 9: w = 512, whose window of 576 slots no other
// register-resident kernel holds for two tasks.
template <int NREG>
__global__ __launch_bounds__(64, NREG <= 3 ? 4 : NREG <= 5 ? 3 : 2) void extz2_pair_mixed_kernel(  // (LDS: ~10 KB a wavefront at two registers -- four to a SIMD at most)
    const PlanTask *__restrict__ plan, const int32_t *__restrict__ order, const uint32_t *__restrict__ pool,
    ScoreK sc, uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  pair_body<NREG, true, false, true>(plan, order, pool, sc, dirbase, res);
}
#define SDF_PAIR_MIXED_INST(N)                                                                                           \
  template __global__ void extz2_pair_mixed_kernel<N>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *, \
                                                      sdf_result *);
SDF_PAIR_MIXED_INST(2)
SDF_PAIR_MIXED_INST(3)
SDF_PAIR_MIXED_INST(4)
SDF_PAIR_MIXED_INST(5)
SDF_PAIR_MIXED_INST(6)
SDF_PAIR_MIXED_INST(8)
SDF_PAIR_MIXED_INST(9)
#undef SDF_PAIR_MIXED_INST

// LDS of a mixed pair: the windows of (max qlen, max tlen) and, behind them, the five state registers of half B per lane
size_t pair_mixed_lds_bytes(int qmax, int tmax, int nreg) {
  return ((2 * (size_t)pair_tcap(tmax, nreg) + 4 * (size_t)pair_qcap(qmax, nreg) + 15) & ~(size_t)15) + (size_t)5 * nreg * 256;
}

// the windows hold the sequences whole?
bool pair_fits_whole(int qlen, int tlen, int nreg) {
  return pair_tcap(tlen, nreg) == (tlen + 15) / 16 * 16 + 64 * nreg + 32 && pair_qcap(qlen, nreg) == qlen + 64 * nreg + 36;
}

size_t pair_lds_bytes(int qlen, int tlen, int nreg) {
  return 2 * (size_t)pair_tcap(tlen, nreg) + 4 * (size_t)pair_qcap(qlen, nreg);
}

}  // namespace sdf
