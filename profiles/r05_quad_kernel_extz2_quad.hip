// Register-resident extz2 DP kernel for gfx950: one wavefront per FOUR DP tasks of equal geometry (round 5).
//
// The pair kernel (extz2_pair.hip) gives two tasks of one (qlen, tlen, w) the 16-bit halves of NREG registers of 64 window
// slots.  For the headline shape (1000 x ~1000, w = 128) that is three registers = 192 slots, of which the reference's
// schedule ever USES 144: re-based at every 16-row block start, the widened band [st & ~15, en | 15] of
// extern/ksw2_extz2_sse.cc:101-115 reaches slot 143 at most, and although the scores are refreshed up to slot 158
// (:124-138) no score written beyond slot 143 is read before it is written again (sdf_plan.hip: quad_window_ok replays
// the schedule of a geometry and checks exactly that).  A quarter of every row's lanes computed nothing.
//
// Here FOUR tasks of one geometry share a wavefront in FIVE registers: pair (A, B) in the halves of physical slots
// 0..159, pair (C, D) in those of slots 160..319 -- 144 window slots + 16 guard slots each; physical slot 64 k + l is
// lane l of register k, so pair two starts at lane 32 of register 2.  1.25 register-rows per task instead of 1.5.
//   * Guard lanes are never written: they keep "never computed" (0 / the wild score), which is also what the
//     (r-1, t-1) shift must feed slot 0 of pair two in the rows between block starts (the plain wave_shr:1 across the
//     seam then needs no fix-up), and what a 16-slot re-base must shift in at a window's top.
//   * Every lane predicate of the row code is a function of the lane's WINDOW slot, the same for both pairs.
//   * Direction flags leave in the pair kernel's layout for three registers (512 B per register and 16-row block and
//     task: slots 144..191 are never written, nor read): the traceback does not know the difference.
//   * The sequences: query windows of both pairs in LDS as in the pair kernel (4 bytes per entry, ready to use); the
//     target codes a lane needs change only when its window is re-based -- every 32 rows -- and come straight from the
//     packed pool then (LDS: 9.6 KB a wavefront, four wavefronts a SIMD).
// Everything else -- the general row for the special cases, the lean rows in their LOW16 / SCALARH / STEADY regimes, the
// exact H along the band's upper edge -- is the pair kernel's, with one more pair.  Not here: streamed windows (long
// sequences), bands that run out (TRACK), mixed pairs; the planner gives this kernel whole-sequence windows of
// 129..144 slots only (sdf_plan.hip).
#include <hip/hip_runtime.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "sdf_internal.h"

namespace sdf {

constexpr int kQuadSlots = 144;   // window slots of a task
constexpr int kQuadStride = 160;  // physical slots of a pair of tasks: the window + 16 guard slots
constexpr int kQuadRegs = 5;

__host__ __device__ inline int quad_qcap(int qlen) { return qlen + kQuadStride + 36; }
// words of one packed target in LDS: 2-bit codes, then the N mask (sdf_pack_codes' layout), rounded to a multiple of four
__host__ __device__ inline int quad_twords(int tlen) { return ((tlen + 15) / 16 + (tlen + 31) / 32 + 3) & ~3; }
size_t quad_lds_bytes(int qlen, int tlen) { return (size_t)2 * 2 * (size_t)((quad_qcap(qlen) + 1) & ~1) + (size_t)16 * quad_twords(tlen); }

// Replays the reference's band schedule for (qlen, tlen, w) with the kernel's re-basing (the window base is the band's
// block-rounded start at every 16th row): true if no cell and no score that is read later ever lies beyond window slot
// 143, and the steady rows' refresh range covers what the row code assumes (slots ra .. ra + 127 at least, ra < 32).
bool quad_window_ok(int qlen, int tlen, int w) {
  if (w < 113 || w > 128 || qlen < 2 * w + 64 || tlen < 2 * w + 64) return false;
  std::vector<int> at(tlen + 160, -1);  // per target position: the slot it had when its score was last refreshed
  std::vector<int> row(tlen + 160, -1);
  int base = 0;
  for (int r = 0; r < qlen + tlen - 1; ++r) {
    Band b;
    if (!band_of(r, qlen, tlen, w, b)) return false;  // (a band that runs out is the TRACK flavour's)
    if ((r & 15) == 0) base = b.lo;
    if (b.hi - base >= kQuadSlots) return false;
    const int nref = ((b.hi0 - b.lo0) & ~15) + 16;
    for (int p = b.lo; p <= b.hi; ++p)
      if (!(p >= b.lo0 && p < b.lo0 + nref) && row[p] >= 0 && at[p] >= kQuadSlots) return false;  // a stale score from a guard slot
    for (int p = b.lo0; p < b.lo0 + nref && p < (int)at.size(); ++p) {
      at[p] = p - base;
      row[p] = r;
    }
  }
  return true;
}

// (the replay costs ~0.3 ms per geometry and the planner asks per chunk and planning thread: answered once per process)
bool quad_window_ok_cached(int qlen, int tlen, int w) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, bool> known;
  const uint64_t key = ((uint64_t)(uint32_t)qlen << 40) | ((uint64_t)(uint32_t)tlen << 16) | (uint64_t)(uint32_t)w;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = known.find(key);
    if (it != known.end()) return it->second;
  }
  const bool ok = quad_window_ok(qlen, tlen, w);
  std::lock_guard<std::mutex> g(mu);
  known[key] = ok;
  return ok;
}

template <bool UNUSED = false>
__device__ __forceinline__ void quad_body(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                          const uint32_t *__restrict__ pool, ScoreK sc, uint8_t *__restrict__ dirbase,
                                          sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NREG = kQuadRegs;
  const PlanTask tk[4] = {plan[order[4 * blockIdx.x]], plan[order[4 * blockIdx.x + 1]], plan[order[4 * blockIdx.x + 2]],
                          plan[order[4 * blockIdx.x + 3]]};  // A, B (pair one: low / high halves), C, D (pair two)
  const int lane = threadIdx.x;
  const int w = tk[0].w, qlen = tk[0].qlen, tlen = tk[0].tlen;  // (one geometry)
  const int nrow = qlen + tlen - 1;
  const int qcap = (quad_qcap(qlen) + 1) & ~1;
  // reversed queries, 32 entries of front pad (extz2_pair.hip), TWO bytes an entry -- first task | second task << 8 --: a
  // byte permute per register-row spreads them over the halves, and the wavefront's LDS is 6 KB instead of 11 (four
  // wavefronts a SIMD: 160 KB / 16)
  uint16_t *W0 = reinterpret_cast<uint16_t *>(lds), *W1 = W0 + qcap;
  // the four packed targets, as they lie in the pool (a lane's target codes change when its window is re-based, every 32
  // rows: unpacked from here then -- a fetch from the pool in HBM at that point stalled the wavefront a microsecond or two
  // per re-base, 10-18 % of its rows' time with three wavefronts a SIMD to hide it)
  const int twords = quad_twords(tlen);
  uint32_t *Tp = reinterpret_cast<uint32_t *>(W1 + qcap);

  // ---- the lane's place: window slot and pair of each of its five registers (guard lanes: slot >= 144) ----
  const bool hi2 = lane >= 32;  // register 2: lanes 0..31 are pair one's slots 128..159, lanes 32..63 pair two's 0..31
  auto slot_of = [&](const int k) -> int {
    return k == 0 ? lane : k == 1 ? 64 + lane : k == 2 ? (hi2 ? lane - 32 : 128 + lane) : k == 3 ? 32 + lane : 96 + lane;
  };
  auto in_slots = [&](const int k, const int a, const int b) -> bool {  // window slots [a, b), never a guard slot
    const int bb = b < kQuadSlots ? b : kQuadSlots;
    return (unsigned)(slot_of(k) - a) < (unsigned)(bb > a ? bb - a : 0);
  };
  // (wave-uniform masks of lane sets that never change; made from lane tests, not literals: see in_mask in extz2_pair.hip)
  const unsigned long long m_valid2 = __ballot(slot_of(2) < kQuadSlots), m_valid4 = __ballot(slot_of(4) < kQuadSlots);
  const unsigned long long m_low0 = __ballot(lane < 16), m_low2 = __ballot(hi2 && lane < 48);  // slots 0..15 of either pair

  // ---- the four queries and the four packed targets into LDS; any N anywhere? ----
  int has_n;
  {
    uint32_t n_seen = 0;
    const int tcw = (tlen + 15) / 16, tnw = (tlen + 31) / 32;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t *tw = pool + tk[t].t_word;
      const uint32_t *qn = pool + tk[t].q_word + (qlen + 15) / 16;
      for (int k = lane; k < tcw + tnw; k += 64) {
        const uint32_t wv = tw[k];
        Tp[t * twords + k] = wv;
        if (k >= tcw) n_seen |= wv;
      }
      for (int k = lane; k < (qlen + 31) / 32; k += 64) n_seen |= qn[k];
    }
    has_n = __builtin_amdgcn_readfirstlane((int)__any(n_seen != 0));
    const uint32_t *qw[4], *qn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qw[t] = pool + tk[t].q_word;
      qn[t] = qw[t] + (qlen + 15) / 16;
    }
    for (int i = lane; i < qcap; i += 64) {
      const int e = i - 32;
      const bool in = e >= 0 && e < qlen;
      W0[i] = in ? (uint16_t)(pool_code8(qw[0], qn[0], qlen - 1 - e, sc.wild) | (pool_code8(qw[1], qn[1], qlen - 1 - e, sc.wild) << 8)) : (uint16_t)0;
      W1[i] = in ? (uint16_t)(pool_code8(qw[2], qn[2], qlen - 1 - e, sc.wild) | (pool_code8(qw[3], qn[3], qlen - 1 - e, sc.wild) << 8)) : (uint16_t)0;
    }
  }
  __syncthreads();
  // target codes of the lane's cell of register k for window base `b`: (code of the pair's first task) | (second) << 16
  auto target_codes = [&](const int k, const int b) -> unsigned {
    const bool second = k > 2 || (k == 2 && hi2);
    const int t = b + slot_of(k);
    unsigned lo = 0u, hi = 0u;
    if (t < tlen) {
      const int tcw = (tlen + 15) / 16;
      const uint32_t *ta = Tp + (second ? 2 : 0) * twords, *tb = ta + twords;
      const uint32_t ca = (ta[t >> 4] >> ((t & 15) * 2)) & 3u, na = (ta[tcw + (t >> 5)] >> (t & 31)) & 1u;
      const uint32_t cb = (tb[t >> 4] >> ((t & 15) * 2)) & 3u, nb = (tb[tcw + (t >> 5)] >> (t & 31)) & 1u;
      lo = na ? (0x80u | sc.wild) : ca;
      hi = nb ? (0x80u | sc.wild) : cb;
    }
    return lo | (hi << 16);
  };

  // ---- constants of the <<8 difference domain (extz2_pair.hip) ----
  const unsigned qb2 = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned qv = qb2;
  const unsigned capv = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned z_match = ((unsigned)((sc.sc_match + sc.qe2_b) & 0xff) << 8) * 0x00010001u;
  const unsigned z_mis_h = ((unsigned)((sc.sc_mis + sc.qe2_b) & 0xff) << 8);
  const unsigned z_delta = ((z_mis_h - (z_match & 0xffffu)) & 0xffffu) * 0x00010001u;
  const unsigned z_wild = ((unsigned)sc.qe2_b << 8) * 0x00010001u;
  unsigned one2 = 0x00010001u;
  asm("" : "+s"(one2));
  unsigned z_match_v = z_match;
  SDF_OPQ(z_match_v);

  unsigned U[NREG], V[NREG], X[NREG], Y[NREG], S[NREG], Tc[NREG];
  unsigned Fa[NREG], Fb[NREG], Fx[NREG], Fy[NREG];
  unsigned xt1[NREG], vt1[NREG];
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
    xt1[k] = vt1[k] = 0u;
    U[k] = V[k] = X[k] = Y[k] = 0u;
    S[k] = z_wild;
    Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    Tc[k] = target_codes(k, 0);
  }

  const bool with_dir = !(tk[0].flag & SDF_FLAG_SCORE_ONLY);
  uint2 *dir[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dir[t] = reinterpret_cast<uint2 *>(dirbase + tk[t].dir_off);

  int base = 0;
  int prev_lo = -1;
  unsigned carry_x[2] = {0u, 0u}, carry_v[2] = {0u, 0u};  // per pair: packed (r-1) values shifted into slot 0 on the first row of a block
  bool zero_low = false;
  int32_t h_top[4] = {0, 0, 0, 0}, h_under[4] = {0, 0, 0, 0};
  int32_t ez_score[4] = {SDF_NEG_INF, SDF_NEG_INF, SDF_NEG_INF, SDF_NEG_INF}, ez_mte[4] = {SDF_NEG_INF, SDF_NEG_INF, SDF_NEG_INF, SDF_NEG_INF};
  int32_t ez_mte_q[4] = {-1, -1, -1, -1};
  int32_t ez_zdropped = 0;
  int drop_row = -1;
  int r0 = 0;
  unsigned qa = 0u, qm = 0u, qc = 0u;  // LDS addresses of the row's query codes: registers 0-1, register 2, registers 3-4
  unsigned hacc[4] = {0u, 0u, 0u, 0u};
  int hcnt = 0;
  auto fold_h = [&]() {
    if (hcnt) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int t = 0; t < 4; ++t) hacc[t] += (unsigned)__shfl_xor((int)hacc[t], off);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        h_under[t] += (int32_t)hacc[t] - hcnt * sc.qe;
        h_top[t] = h_under[t];
        hacc[t] = 0u;
      }
      hcnt = 0;
    }
  };
  // a lane of a window slot, by pair: value of register array A at window slot s of pair p (wave-uniform s)
#define SDF_QREAD(A, p, s)                                                                                   \
  ((p) == 0 ? ((s) < 64 ? (unsigned)__builtin_amdgcn_readlane((int)A[0], (s)&63)                               \
                        : (s) < 128 ? (unsigned)__builtin_amdgcn_readlane((int)A[1], (s)&63)                   \
                                    : (unsigned)__builtin_amdgcn_readlane((int)A[2], (s)&63))                  \
            : ((s) < 32 ? (unsigned)__builtin_amdgcn_readlane((int)A[2], ((s) + 32) & 63)                      \
                        : (s) < 96 ? (unsigned)__builtin_amdgcn_readlane((int)A[3], ((s)-32) & 63)             \
                                   : (unsigned)__builtin_amdgcn_readlane((int)A[4], ((s)-96) & 63)))
  auto h_row = [&](const int r, const int hi0, const int lo0, const int hi, const bool first, const bool want_top,
                   const bool top_from_under, const bool up, const unsigned (&uh)[2], const unsigned (&vu)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int32_t uht = (int32_t)((uh[t >> 1] >> (16 * (t & 1) + 8)) & 0xffu), vut = (int32_t)((vu[t >> 1] >> (16 * (t & 1) + 8)) & 0xffu);
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
  // (r-1, t-1) neighbours of a general row: x, v up by one slot, the pairs' slot 0 from their carries
  auto shift_general = [&](const unsigned (&xcarry)[2], const unsigned (&vcarry)[2]) {
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      if (k == 0) {
        xt1[0] = (unsigned)__builtin_amdgcn_update_dpp((int)xcarry[0], (int)X[0], 0x138, 0xf, 0xf, false);
        vt1[0] = (unsigned)__builtin_amdgcn_update_dpp((int)vcarry[0], (int)V[0], 0x138, 0xf, 0xf, false);
      } else {
        const int x0 = __builtin_amdgcn_mov_dpp((int)X[k - 1], 0x13C, 0x1, 0x1, false);
        xt1[k] = (unsigned)__builtin_amdgcn_update_dpp(x0, (int)X[k], 0x138, 0xf, 0xf, false);
        const int v0 = __builtin_amdgcn_mov_dpp((int)V[k - 1], 0x13C, 0x1, 0x1, false);
        vt1[k] = (unsigned)__builtin_amdgcn_update_dpp(v0, (int)V[k], 0x138, 0xf, 0xf, false);
      }
    }
    if (lane == 32) {  // pair two's slot 0
      xt1[2] = xcarry[1];
      vt1[2] = vcarry[1];
    }
  };

  // ------------------------------------------------------------------------------------------
  // General row (extz2_pair.hip: slow_row): every special case of the reference.
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
    const bool ref_rebased = lo != prev_lo && prev_lo >= 0;
    if (ref_rebased && off_lo == 16) drop_row = r;
    if (off_lo == 16 && !ref_rebased && !zero_low) {
      if (in_mask(m_low0)) {
        X[0] = 0u;
        V[0] = 0u;
      }
      if (in_mask(m_low2)) {
        X[2] = 0u;
        V[2] = 0u;
      }
      zero_low = true;
    }
    // ---- boundary cell t = r: y = 0, u = gap open (reference :122) ----
    if (hi >= r) {
      const int sr = r - base;
      const unsigned uval = r ? qb2 : 0u;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        const bool mine = slot_of(k) == sr;
        U[k] = mine ? uval : U[k];
        Y[k] = mine ? 0u : Y[k];
      }
    }
    // ---- (r-1, t-1) neighbours ----
    {
      unsigned vcarry[2], xcarry[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        vcarry[p] = (base == 0 && r > 0) ? qb2 : (r == r0 ? carry_v[p] : 0u);
        xcarry[p] = (base != 0 && r == r0) ? carry_x[p] : 0u;
      }
      shift_general(xcarry, vcarry);
      // sign-extension artefact of the reference's carry-in (:145-146), either pair
      if (ref_rebased) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (off_lo == 16) {
            const unsigned sm = sign_smear(SDF_QREAD(V, p, 15));
            if (sm && (p == 0 ? in_slots(0, 17, 20) : in_slots(2, 17, 20) && hi2)) vt1[p == 0 ? 0 : 2] |= sm;
          } else if (r == r0) {
            const unsigned sm = sign_smear(carry_v[p]);
            if (sm && (p == 0 ? in_slots(0, 1, 4) : in_slots(2, 1, 4) && hi2)) vt1[p == 0 ? 0 : 2] |= sm;
          }
        }
      }
    }
    // ---- scores: refresh [lo0, lo0 + 16 n), keep the old value elsewhere ----
    {
      const int ra = lo0 - base;
      const int rb = ra + ((hi0 - lo0) & ~15) + 16;
      const int cq = qlen - 1 - r + base + 32;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        const int s = slot_of(k);
        const unsigned q16 = (k > 2 || (k == 2 && hi2)) ? W1[cq + s] : W0[cq + s];
        const unsigned qcw = __builtin_amdgcn_perm(0u, q16, 0x0c010c00u);
        unsigned z;
        SDF_PFRESH(z, Tc[k], qcw, has_n)
        if (in_slots(k, ra, rb)) S[k] = z;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- the recurrence on the reference's widened range [lo, hi] ----
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      if (in_slots(k, off_lo, off_hi + 1)) SDF_CORE(k)
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- exact H of the top cell and of the cell under the band edge (score, mte) ----
    {
      const int st = hi0 - base;
      unsigned uh[2] = {0u, 0u}, vu[2] = {0u, 0u};
      int hin = (r + 1 + w) >> 1;
      hin = hin > r + 1 ? r + 1 : hin;
      hin = hin > tlen - 1 ? tlen - 1 : hin;
      const bool up = hin == hi0 + 1 || hin == 0;
      const bool want_top = up || hi0 == tlen - 1 || hi0 == 0;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (want_top) {
          const unsigned ru = SDF_QREAD(U, p, st), rv = SDF_QREAD(V, p, st);
          uh[p] = hi0 > 0 ? ru : rv;
        }
        if (!up && st > 0) vu[p] = SDF_QREAD(V, p, st - 1);
      }
      h_row(r, hi0, lo0, hi, r == 0, want_top, hi0 > 0, up, uh, vu);
    }
    prev_lo = lo;
    return true;
  };

  // ------------------------------------------------------------------------------------------
  // Lean rows [rb, re) of one block (extz2_pair.hip: lean_rows_n).  STEADY: pure band regime, base >= 16, the refresh range
  // covers slots ra .. ra + 127 at least with ra < 32 -- registers 1 and 3 whole, 0 / 2 / 4 by scalar masks.
  // ------------------------------------------------------------------------------------------
  auto lean_rows_n = [&](auto low16_c, auto scalarh_c, auto steady_c, auto hasn_c, const int rb, const int re) {
    constexpr bool HASN = decltype(hasn_c)::value;
    constexpr bool LOW16 = decltype(low16_c)::value;
    constexpr bool SCALARH = decltype(scalarh_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    if (SCALARH) fold_h();
    {
      const unsigned e0 = (unsigned)(2 * (qlen - 1 - rb + base + 32));
      qa = e0 + 2u * lane;                                                    // pair one, slots 0.. (register k: + 128 k)
      qc = (unsigned)(2 * qcap) + e0 + 2u * (32 + lane);                       // pair two, slots 32.. (register 3; register 4: + 128)
      qm = hi2 ? (unsigned)(2 * qcap) + e0 + 2u * (lane - 32) : qa + 256u;      // register 2
    }
    if (STEADY && !SCALARH) hcnt += re - rb;
    const unsigned vcar = base == 0 ? qb2 : 0u;
    // lanes of registers 0 / 2 / 4 the recurrence runs on
    const unsigned long long m_core0 = ~m_low0, m_core2 = LOW16 ? (m_valid2 & ~m_low2) : m_valid2;  // (m_core0: LOW16 only)
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
      unsigned qcur[NREG];  // (two bytes an entry -> the halves of a register)
      qcur[0] = __builtin_amdgcn_perm(0u, (unsigned)*reinterpret_cast<const uint16_t *>(lds + qa), 0x0c010c00u);
      qcur[1] = __builtin_amdgcn_perm(0u, (unsigned)*reinterpret_cast<const uint16_t *>(lds + qa + 128), 0x0c010c00u);
      qcur[2] = __builtin_amdgcn_perm(0u, (unsigned)*reinterpret_cast<const uint16_t *>(lds + qm), 0x0c010c00u);
      qcur[3] = __builtin_amdgcn_perm(0u, (unsigned)*reinterpret_cast<const uint16_t *>(lds + qc), 0x0c010c00u);
      qcur[4] = __builtin_amdgcn_perm(0u, (unsigned)*reinterpret_cast<const uint16_t *>(lds + qc + 128), 0x0c010c00u);
      qa -= 2;
      qm -= 2;
      qc -= 2;
      // boundary cell t = r (reference :122)
      if (!STEADY && off_hi + base >= r) {
        const int sr = r - base;
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          const bool mine = slot_of(k) == sr;
          U[k] = mine ? qb2 : U[k];
          Y[k] = mine ? 0u : Y[k];
        }
      }
      // scores: refreshed slots are [ra, rbe)
      const int ra = lo0 - base;
      const int rbe = ra + ((hi0 - lo0) & ~15) + 16;
      const int top = rbe < kQuadSlots ? rbe : kQuadSlots;  // (scores beyond the window are never read: quad_window_ok)
      // (steady rows: 0 <= ra < 32 and 128 <= top <= 144, so the masks are plain shifts -- a handful of scalar instructions
      // instead of the guarded general form)
      const unsigned long long m0 = STEADY ? (~0ull << ra) : 0ull;
      const unsigned long long m2 = STEADY ? (((1ull << (top - 128)) - 1ull) | (~0ull << (32 + ra))) : 0ull;
      const unsigned long long m4 = STEADY ? ((1ull << (top - 96)) - 1ull) : 0ull;
      // Register by register from the TOP one down: the (r-1, t-1) shift of register k takes lane 0 from register k - 1 as
      // the row before left it, which is still there when k is done first -- so the shifted x, v are temporaries of one
      // register's step instead of ten registers kept over the row.
#pragma unroll
      for (int k = NREG - 1; k >= 0; --k) {
        if (k == 0) {
          xt1[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X[0], 0x138, 0xf, 0xf, true);
          if (STEADY) vt1[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)V[0], 0x138, 0xf, 0xf, true);
          else vt1[0] = (unsigned)__builtin_amdgcn_update_dpp((int)vcar, (int)V[0], 0x138, 0xf, 0xf, false);
        } else {
          const int x0 = __builtin_amdgcn_mov_dpp((int)X[k - 1], 0x13C, 0x1, 0x1, false);
          xt1[k] = (unsigned)__builtin_amdgcn_update_dpp(x0, (int)X[k], 0x138, 0xf, 0xf, false);
          const int v0 = __builtin_amdgcn_mov_dpp((int)V[k - 1], 0x13C, 0x1, 0x1, false);
          vt1[k] = (unsigned)__builtin_amdgcn_update_dpp(v0, (int)V[k], 0x138, 0xf, 0xf, false);
        }
        // (pair two's slot 0 took lane 31 of register 2: a guard lane, x = v = 0 -- what a row between block starts feeds
        // slot 0 when the window does not start at target position 0; when it does, v is the gap open)
        if (k == 2 && !STEADY && base == 0 && lane == 32) vt1[2] = vcar;
        unsigned z;
        SDF_PFRESH(z, Tc[k], qcur[k], HASN)
        if (STEADY) {
          if (k == 0) S[0] = in_mask(m0) ? z : S[0];
          else if (k == 2) S[2] = in_mask(m2) ? z : S[2];
          else if (k == 4) S[4] = in_mask(m4) ? z : S[4];
          else S[k] = z;
          if (k == 0) {
            if (LOW16) {
              if (in_mask(m_core0)) SDF_CORE(0)
            } else {
              SDF_CORE(0)
            }
          } else if (k == 2) {
            if (in_mask(m_core2)) SDF_CORE(2)
          } else if (k == 4) {
            if (in_mask(m_valid4)) SDF_CORE(4)
          } else {
            SDF_CORE(k)
          }
        } else {
          if (in_slots(k, ra, rbe)) S[k] = z;
          if (in_slots(k, LOW16 ? 16 : 0, off_hi + 1)) SDF_CORE(k)
        }
      }
      if (SCALARH) {
        const int st = hi0 - base;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const unsigned uh = SDF_QREAD(U, p, st), vu = st > 0 ? SDF_QREAD(V, p, st - 1) : 0u;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int t = 2 * p + h;
            h_top[t] = h_under[t] + (int32_t)((uh >> (16 * h + 8)) & 0xffu) - sc.qe;
            if (hi0 - 1 >= lo0) h_under[t] += (int32_t)((vu >> (16 * h + 8)) & 0xffu) - sc.qe;
            if (h_top[t] > ez_mte[t]) {
              ez_mte[t] = h_top[t];
              ez_mte_q[t] = r - (hi0 | 15);
            }
            if (r == nrow - 1) ez_score[t] = h_top[t];
          }
        }
      } else {
        // H path (extz2_pair.hip): the lane that owns the path's cell adds its u or v; one lane per pair
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
          if (STEADY) {  // (sl >= 126: pair one in register 1 or 2, pair two in register 4)
            const unsigned u1 = U[1], v1 = V[1], u2 = U[2], v2 = V[2], u4 = U[4], v4 = V[4];
            const unsigned c1 = up ? u1 : v1, c2 = up ? u2 : v2, c4 = up ? u4 : v4;
            const unsigned val0 = sl < 128 ? c1 : c2;
            if (in_mask(1ull << (sl & 63))) {
              hacc[0] += (val0 >> 8) & 0xffu;
              hacc[1] += val0 >> 24;
            }
            if (in_mask(1ull << ((sl - 96) & 63))) {
              hacc[2] += (c4 >> 8) & 0xffu;
              hacc[3] += c4 >> 24;
            }
          } else {
#pragma unroll
            for (int k = 0; k < NREG; ++k) {
              const unsigned ck = up ? U[k] : V[k];
              if (slot_of(k) == sl) {  // (one lane of one register per pair)
                if (k < 2 || (k == 2 && !hi2)) {
                  hacc[0] += (ck >> 8) & 0xffu;
                  hacc[1] += ck >> 24;
                } else {
                  hacc[2] += (ck >> 8) & 0xffu;
                  hacc[3] += ck >> 24;
                }
              }
            }
            ++hcnt;
          }
        }
      }
    }
  };
  auto lean_rows = [&](auto low16_c, auto scalarh_c, auto steady_c, const int rb, const int re) {
    if (has_n) lean_rows_n(low16_c, scalarh_c, steady_c, std::true_type{}, rb, re);
    else lean_rows_n(low16_c, scalarh_c, steady_c, std::false_type{}, rb, re);
  };
  auto zero_cells = [&](const int t_from, const int t_to) {
#pragma unroll
    for (int k = 0; k < NREG; ++k)
      if (in_slots(k, t_from - base, t_to - base + 1)) {
        U[k] = 0u;
        V[k] = 0u;
        X[k] = 0u;
        Y[k] = 0u;
      }
  };
  int win_hi = -1;
  int dirty_hi = -1;

  for (r0 = 0; r0 < nrow && !ez_zdropped; r0 += 16) {
    // ---- block start: re-base the windows to the reference's band start of this row ----
    {
      Band b0;
      if (!band_of(r0, qlen, tlen, w, b0)) {
        ez_zdropped = 1;
        break;
      }
      carry_x[0] = carry_x[1] = carry_v[0] = carry_v[1] = 0u;
      if (b0.lo != base) {  // always +16: every window 16 slots down
        if (prev_lo == base) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            carry_x[p] = SDF_QREAD(X, p, 15);
            carry_v[p] = SDF_QREAD(V, p, 15);
          }
        }
        const int bidx = ((lane + 16) & 63) * 4;
        const bool from_next = lane >= 48;
        const bool guard2 = !hi2 && lane >= 16;  // pair one's guard slots: the shift put pair two's slots 0..15 there
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          unsigned a0, a1;
#define SDF_QSHIFT16(A, INIT)                                                            \
  a0 = (unsigned)__builtin_amdgcn_ds_bpermute(bidx, (int)A[k]);                         \
  a1 = (k + 1 < NREG) ? (unsigned)__builtin_amdgcn_ds_bpermute(bidx, (int)A[k + 1 < NREG ? k + 1 : k]) : (INIT); \
  A[k] = from_next ? a1 : a0;                                                           \
  if (k == 2) A[2] = guard2 ? (INIT) : A[2];                                            \
  __builtin_amdgcn_sched_barrier(0);
          SDF_QSHIFT16(U, 0u)
          SDF_QSHIFT16(V, 0u)
          SDF_QSHIFT16(X, 0u)
          SDF_QSHIFT16(Y, 0u)
          SDF_QSHIFT16(S, z_wild)
#undef SDF_QSHIFT16
        }
        base = b0.lo;
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          Tc[k] = target_codes(k, base);
          __builtin_amdgcn_sched_barrier(0);
        }
        zero_low = false;
      }
    }
    const int rend = r0 + 16 < nrow ? r0 + 16 : nrow;
    drop_row = -1;
    int r = r0;
    {
      const int rl = r0 + 15;
      bool steady = w >= 2 && r0 + 16 <= nrow && base >= 16 && ((rl - w + 1) >> 1) >= rl - qlen + 1 &&
                    ((rl + w) >> 1) < tlen - 1 && ((r0 + w) >> 1) + 15 < r0;
      if (steady) {
        // refresh range [ra, ra + 128 or more) with ra < 32 on all sixteen rows; the H path's cell at slot 126 or beyond
        const int lo0a = (r0 - w + 1) >> 1, hi0a = (r0 + w) >> 1;
        const int ra_last = ((rl - w + 1) >> 1) - base;
        steady = ((w - 1) & ~15) + 16 >= 128 && ra_last < 32 && lo0a - base >= 0 && hi0a - 1 - base >= 126 &&
                 (((rl + w) >> 1) | 15) - base < kQuadSlots;
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
        if (hi > win_hi) {
          if (dirty_hi > win_hi) zero_cells(win_hi + 1, hi < dirty_hi ? hi : dirty_hi);
          win_hi = hi;
          if (dirty_hi < win_hi) dirty_hi = win_hi;
        }
        const bool rebase_row = lo != prev_lo && prev_lo >= 0;
        bool special = !lean_ok || r == 0 || (r == r0 && (carry_x[0] | carry_v[0] | carry_x[1] | carry_v[1]) != 0u);
        if (rebase_row && !special) {
          const unsigned cv0 = lo - base == 16 ? SDF_QREAD(V, 0, 15) : carry_v[0], cv1 = lo - base == 16 ? SDF_QREAD(V, 1, 15) : carry_v[1];
          special = ((cv0 | cv1) & 0x80008000u) != 0u;
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
          if (in_mask(m_low0)) {
            X[0] = 0u;
            V[0] = 0u;
          }
          if (in_mask(m_low2)) {
            X[2] = 0u;
            V[2] = 0u;
          }
          zero_low = true;
        }
        int stop = rend;
        if (rebase_row) {
          stop = r + 1;
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
        // (the steady rows run every window lane: slots above the reference's window hold scratch values afterwards; the
        // other lean rows mask their lanes to the window)
        if (steady && !scalarh && base + kQuadSlots - 1 > dirty_hi) dirty_hi = base + kQuadSlots - 1;
        prev_lo = lo;
        r = stop;
      }
      if (dirty_hi > win_hi) zero_cells(win_hi + 1, dirty_hi);  // clean lanes for the re-base shift
      dirty_hi = win_hi;
    }
    // ---- block end: direction flags of these (<= 16) rows leave for HBM, in the pair kernel's three-register layout ----
    if (with_dir) {
      const int done = r - r0;
      const int rbk = r0 >> 4;
      if (drop_row >= 0) {
        const unsigned sh = (unsigned)(r - drop_row);
        if (in_mask(m_low0)) {
          Fa[0] = pk_shl(Fa[0], sh);
          Fb[0] = pk_shl(Fb[0], sh);
          Fx[0] = pk_shl(Fx[0], sh);
          Fy[0] = pk_shl(Fy[0], sh);
        }
        if (in_mask(m_low2)) {
          Fa[2] = pk_shl(Fa[2], sh);
          Fb[2] = pk_shl(Fb[2], sh);
          Fx[2] = pk_shl(Fx[2], sh);
          Fy[2] = pk_shl(Fy[2], sh);
        }
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
          const int s = slot_of(k);
          if (s < kQuadSlots) {
            const bool second = k > 2 || (k == 2 && hi2);
            const int64_t at = ((int64_t)rbk * 3 + (s >> 6)) * 64 + (s & 63);
            uint2 *d0 = second ? dir[2] : dir[0], *d1 = second ? dir[3] : dir[1];
            d0[at] = make_uint2(__builtin_amdgcn_perm(fb, fa, 0x05040100u), __builtin_amdgcn_perm(fy, fx, 0x05040100u));
            d1[at] = make_uint2(__builtin_amdgcn_perm(fb, fa, 0x07060302u), __builtin_amdgcn_perm(fy, fx, 0x07060302u));
          }
          __builtin_amdgcn_sched_barrier(0);  // (one register's stores at a time: the cold code must not cost the rows their registers)
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
  }

  fold_h();
  if (lane < 4) {
    sdf_result o;
    o.score = lane == 0 ? ez_score[0] : lane == 1 ? ez_score[1] : lane == 2 ? ez_score[2] : ez_score[3];
    o.max = 0;
    o.max_q = -1;
    o.max_t = -1;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = lane == 0 ? ez_mte[0] : lane == 1 ? ez_mte[1] : lane == 2 ? ez_mte[2] : ez_mte[3];
    o.mte_q = lane == 0 ? ez_mte_q[0] : lane == 1 ? ez_mte_q[1] : lane == 2 ? ez_mte_q[2] : ez_mte_q[3];
    o.zdropped = ez_zdropped;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[lane == 0 ? tk[0].out_idx : lane == 1 ? tk[1].out_idx : lane == 2 ? tk[2].out_idx : tk[3].out_idx] = o;
  }
#undef SDF_QREAD
}

__global__ __launch_bounds__(64, 3) void extz2_quad_kernel(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                                           const uint32_t *__restrict__ pool, ScoreK sc,
                                                           uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  quad_body<>(plan, order, pool, sc, dirbase, res);
}

}  // namespace sdf
