// Register-resident extz2 DP for targets wider than one wavefront's window: a workgroup of up to 16 wavefronts
// works on ONE full-band task as a systolic array.
//
// The target is cut into stripes of NSLOT = 128*NREG positions, one wavefront each.  In stripe-local
// coordinates (t' = t - T0, r' = r - T0) the reference's band of a full-band task (lo = max(0, r-qlen+1),
// hi = min(r, tlen-1), extern/ksw2_extz2_sse.cc:101-115) has the same form as that of a stand-alone task over
// the target slice -- so each wavefront runs the one-task kernel of extz2_wave.hip on its slice, with three
// differences at the stripe's left edge:
//   * the (r-1, t-1) neighbour of its first column is the left stripe's last column of the previous global row
//     (x, v), not the reference's start-of-target constants: stripes export that column row by row into an LDS
//     ring and publish their progress once per 16-row block; a stripe waits for its left neighbour's block
//     before its own (and a producer never runs more than a ring ahead of its consumer);
//   * "first row" rules (r == 0) apply to the global row; the exact H of the top cell, which moves up one
//     column per row until it reaches the end of the target, is handed from stripe to stripe;
//   * only the last stripe owns the target's end: score / mte come from it.
// Wavefront s starts NSLOT rows after wavefront s-1 and all of them advance together afterwards: a 6000 x 6000
// task keeps 12 wavefronts of a CU busy instead of one workgroup stepping through LDS-resident state.
// Direction flags: the wave-kernel bit blocks, one region per stripe (traceback layout 3).
//
// Compiled inside sdf_unity.hip after extz2_wave.hip (helpers, SDF_CORE, SDF_FRESH, slot_half, sel*).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "sdf_internal.h"

namespace sdf {

#define SDF_RING 256  // rows of edge values a stripe may run ahead of its right neighbour

template <int NREG>
__global__ __launch_bounds__(1024) void extz2_stripe_kernel(const PlanTask *__restrict__ plan,
                                                                              const int32_t *__restrict__ order,
                                                                              const uint32_t *__restrict__ pool, ScoreK sc,
                                                                              uint8_t *__restrict__ dirbase,
                                                                              sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NSLOT = 128 * NREG;  // window slots = stripe width
  const PlanTask tk = plan[order[blockIdx.x]];
  const int lane = threadIdx.x & 63;
  const int sb = threadIdx.x >> 6;  // stripe of this wavefront
  const int tlen_all = tk.tlen;
  const int nstripe = (tlen_all + NSLOT - 1) / NSLOT;
  const int qlen = tk.qlen, w = tk.w;
  const int T0 = sb * NSLOT;                                           // first target position of the stripe
  const int tlen = tlen_all - T0 < NSLOT ? tlen_all - T0 : NSLOT;      // its slice (<= 0: no such stripe)
  // LDS: sync words | rings | reversed query (shared) | target slices (one per stripe)
  volatile int *prog_prod = reinterpret_cast<volatile int *>(lds);       // [16] last global row stripe s has completed
  volatile int *prog_cons = prog_prod + 16;  // [16] last global row whose edge (of stripe s) stripe s+1 is done with
  volatile int *hand_val = prog_prod + 32;   // [16] H of the top cell after stripe s's row NSLOT-1
  volatile uint32_t *rings = reinterpret_cast<volatile uint32_t *>(lds + 256);  // [16][SDF_RING] x | v << 16
  const int qcap = qlen + NSLOT + 36, tcap = 2 * NSLOT + 32;
  const int wofs = 256 + 16 * SDF_RING * 4;  // byte offset of the shared reversed query
  uint16_t *W = reinterpret_cast<uint16_t *>(lds + wofs);
  uint16_t *Tb = reinterpret_cast<uint16_t *>(lds + wofs + ((2 * qcap + 15) & ~15)) + sb * tcap;
  const int64_t tw_off = tk.t_word, qw_off = tk.q_word;
#define tt0 0
#define we0 0

  // ---- unpack: the query once for the workgroup, each wavefront its slice of the target ----
  bool has_n;
  {
    const uint32_t *tw = pool + tw_off, *tn = tw + (tlen_all + 15) / 16;
    const uint32_t *qw = pool + qw_off, *qn = qw + (qlen + 15) / 16;
    uint32_t n_seen = 0;
    for (int k = lane; k < (tlen_all + 31) / 32; k += 64) n_seen |= tn[k];
    for (int k = lane; k < (qlen + 31) / 32; k += 64) n_seen |= qn[k];
    has_n = __builtin_amdgcn_readfirstlane((int)__any(n_seen != 0)) != 0;  // wave-uniform
    if (threadIdx.x < 16) {
      prog_prod[threadIdx.x] = -1;
      // nothing of an edge is needed before the next stripe's first row; no next stripe: never wait
      prog_cons[threadIdx.x] = (int)threadIdx.x + 1 < nstripe ? ((int)threadIdx.x + 1) * NSLOT - 1 : 0x7fffffff;
      hand_val[threadIdx.x] = 0;
    }
    if (sb < nstripe)  // (wavefronts beyond the task's last stripe have no slice, nor LDS for one)
      for (int i = lane; i < tcap; i += 64) Tb[i] = i < tlen ? (uint16_t)pool_code16(tw, tn, T0 + i, sc.wild) : 0;
    for (int i = threadIdx.x; i < qcap; i += blockDim.x) {
      const int e0 = i - 32, e1 = e0 + 1;  // QR indices; QR[e] = query[qlen-1-e], 0 outside
      uint32_t v0 = (e0 >= 0 && e0 < qlen) ? pool_code16(qw, qn, qlen - 1 - e0, sc.wild) : 0u;
      uint32_t v1 = (e1 >= 0 && e1 < qlen) ? pool_code16(qw, qn, qlen - 1 - e1, sc.wild) : 0u;
      v0 = (v0 & 0x7fu) | ((v0 >> 8) & 0x80u);
      v1 = (v1 & 0x7fu) | ((v1 >> 8) & 0x80u);
      W[i] = (uint16_t)(v0 | (v1 << 8));
    }
  }
  __syncthreads();
  if (sb >= nstripe) return;
  const bool has_left = sb > 0, has_right = sb + 1 < nstripe;
  // the right neighbour reads this stripe's edge while its own window starts at its first column
  const int export_until = has_right ? qlen + NSLOT + 32 : 0;
  volatile uint32_t *ring_out = rings + sb * SDF_RING, *ring_in = rings + (sb > 0 ? sb - 1 : 0) * SDF_RING;
  auto wait_ge = [&](volatile int *p, const int need) {
    while (*p < need) __builtin_amdgcn_s_sleep(2);
    __threadfence_block();
  };

  // ---- constants of the <<8 difference domain ----
  const unsigned qv = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned capv = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned z_match = ((unsigned)((sc.sc_match + sc.qe2_b) & 0xff) << 8) * 0x00010001u;
  const unsigned z_mis_h = ((unsigned)((sc.sc_mis + sc.qe2_b) & 0xff) << 8);
  const unsigned z_delta = ((z_mis_h - (z_match & 0xffffu)) & 0xffffu) * 0x00010001u;
  const unsigned z_wild = ((unsigned)sc.qe2_b << 8) * 0x00010001u;  // score 0, also "never written"
  unsigned one2 = 0x00010001u;  // min(x, 1) per half; opaque so that it stays one v_pk_min_u16
  SDF_OPQ(one2);
  unsigned z_match_v = z_match;  // kept in a VGPR: v_pk_mad_u16 takes one scalar operand only
  SDF_OPQ(z_match_v);

  unsigned U[NREG], V[NREG], X[NREG], Y[NREG], S[NREG], Tc[NREG];
  unsigned Fa[NREG], Fb[NREG], Fx[NREG], Fy[NREG];
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
    U[k] = V[k] = X[k] = Y[k] = 0u;
    S[k] = z_wild;
    Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    Tc[k] = *reinterpret_cast<const uint32_t *>(Tb + 128 * k + 2 * lane);
  }

  const bool with_dir = !(tk.flag & SDF_FLAG_SCORE_ONLY);
  // flags: one region per stripe, sized for a full stripe
  uint4 *dir = reinterpret_cast<uint4 *>(dirbase + tk.dir_off + (int64_t)sb * ((int64_t)((qlen + NSLOT - 1 + 15) / 16) * NREG * 1024));
  const int nrow = qlen + tlen - 1;
  const int bperm_idx = ((lane + 8) & 63) * 4;

  int base = 0;
  int prev_lo = -1;
  unsigned carry_x = 0u, carry_v = 0u;  // halves shifted into slot 0 on the first row of a block
  bool zero_low = false;  // slots below the reference window still hold x,v that must read as 0
  int32_t h_top = 0, h_under = 0;  // H of the top cell / of the cell the next top cell will read
  bool track_h = true;  // (a stripe hands the top cell's H to the next one after its row NSLOT-1)
  if (has_left) {  // the left stripe has finished global row T0-1: its edge values and the H of the top cell are there
    wait_ge(prog_prod + sb - 1, T0 - 1);
    h_top = h_under = hand_val[sb - 1];
  }
  int32_t ez_score = SDF_NEG_INF, ez_mte = SDF_NEG_INF, ez_mte_q = -1, ez_zdropped = 0;
  int drop_row = -1;  // row of the current block at which the reference window left slots 0..15
  int r0 = 0;
  unsigned qaddr = 0u, qnext[NREG];  // LDS address / prefetched query codes of row `qrow` (lean rows)
  int qrow = -1;
#pragma unroll
  for (int k = 0; k < NREG; ++k) qnext[k] = 0u;
  unsigned hacc = 0u;  // lane-distributed part of the H path sum (lean rows), folded lazily
  int hcnt = 0;        // number of path steps in hacc (each subtracts q+e)
  auto fold_h = [&]() {  // bring the scalar path value up to date
    if (hcnt) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) hacc += (unsigned)__shfl_xor((int)hacc, off);
      h_under += (int32_t)hacc - hcnt * sc.qe;
      h_top = h_under;
      hacc = 0u;
      hcnt = 0;
    }
  };

  // the stripe's last column (x, v of target position T0 + NSLOT - 1 after row r) for the right neighbour
  auto export_edge = [&](const int r) {
    const int s = NSLOT - 1 - base;  // its slot
    unsigned xe = 0u, ve = 0u;
#pragma unroll
    for (int k = 0; k < NREG; ++k)
      if ((s >> 7) == k) {
        xe = slot_half(X[k], s & 127);
        ve = slot_half(V[k], s & 127);
      }
    if (lane == 0) ring_out[(r + T0) & (SDF_RING - 1)] = xe | (ve << 16);
  };

  // ------------------------------------------------------------------------------------------
  // General row: every special case of the reference (first/last rows, boundary cell t = r,
  // clipping by the sequence ends, carry-in artefacts).  Returns false when the band is exhausted.
  // ------------------------------------------------------------------------------------------
  auto slow_row = [&](const int r) -> bool {
    fold_h();
    // band of this row (reference :101-115); eligibility guarantees it is never empty
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
      if (lane < 8) {
        X[0] = 0u;
        V[0] = 0u;
      }
      zero_low = true;
    }
    // ---- boundary cell t = r: y = 0, u = gap open (reference :122) ----
    if (hi >= r) {
      const int sr = r - base;
      const unsigned keep = (sr & 1) ? 0x0000ffffu : 0xffff0000u;
      const unsigned uval = (r + T0) ? (((unsigned)sc.q_b << 8) << ((sr & 1) * 16)) : 0u;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {  // selects on every register: conditional stores into the arrays would be
        const bool mine = (sr >> 7) == k && lane == ((sr & 127) >> 1);  // merged into a dynamically indexed store
        U[k] = mine ? ((U[k] & keep) | uval) : U[k];
        Y[k] = mine ? (Y[k] & keep) : Y[k];
      }
    }
    // ---- (r-1, t-1) neighbours: shift x and v up by one slot ----
    unsigned xt1[NREG], vt1[NREG];
    {
      // carry into slot 0: x = 0, v = gap open when the window starts at t = 0 (r > 0); the
      // captured (r-1) values when the reference re-bases exactly at a block start
      // (a stripe with a left neighbour: the neighbour's edge values of the previous global row instead)
      const bool feed = has_left && base == 0;
      const uint32_t fe = feed ? ring_in[(r + T0 - 1) & (SDF_RING - 1)] : 0u;
      const unsigned vcarry = feed ? (fe & 0xffff0000u)
                              : (base == 0 && r > 0) ? ((unsigned)sc.q_b << 24) : (r == r0 ? carry_v << 16 : 0u);
      const unsigned xcarry = feed ? fe << 16 : (base != 0 && r == r0) ? carry_x << 16 : 0u;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        unsigned xs, vs;
        if (k == 0) {
          xs = (unsigned)__builtin_amdgcn_update_dpp((int)xcarry, (int)X[0], 0x138, 0xf, 0xf, false);
          vs = (unsigned)__builtin_amdgcn_update_dpp((int)vcarry, (int)V[0], 0x138, 0xf, 0xf, false);
        } else {
          int ux, uv;
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
      // sign-extension artefact of the reference's carry-in (:145-146): a negative v carry also
      // sets lanes 1..3 of the first block.  Only possible on the reference's rebase rows.
      if (ref_rebased) {
        if (off_lo == 16) {
          const unsigned cvh = slot_half(V[0], 15);
          if (cvh & 0x8000u) {
            if (lane == 8) vt1[0] |= 0xff000000u;
            if (lane == 9) vt1[0] |= 0xff00ff00u;
          }
        } else if (r == r0 && (carry_v & 0x8000u)) {
          if (lane == 0) vt1[0] |= 0xff000000u;
          if (lane == 1) vt1[0] |= 0xff00ff00u;
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
        const int a_ = ra - 128 * k, b_ = rb - 128 * k;
        if (b_ > 0 && a_ < 128) {
          const unsigned qc = W[cq - we0 + 128 * k + 2 * lane];  // zero-extended byte pair
          unsigned z;
          SDF_FRESH(z, Tc[k], qc)
          if (a_ <= 0 && b_ >= 128) {
            S[k] = z;
          } else {
            sel_lo_rng(S[k], z, (a_ + 1) >> 1, (b_ + 1) >> 1, lane);
            sel_hi_rng(S[k], z, a_ >> 1, b_ >> 1, lane);
          }
        }
      }
    }
    // ---- the recurrence on the reference's widened range [lo, hi] ----
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int l0 = off_lo - 128 * k <= 0 ? 0 : (off_lo - 128 * k) >> 1;
      const int l1 = (off_hi - 128 * k) >> 1;  // off_hi is odd
      if (l1 >= l0 && l0 < 64) {
        if ((unsigned)(lane - l0) <= (unsigned)(l1 - l0)) SDF_CORE(k)
      }
    }
    // ---- exact H of the top cell and of the cell under the band edge (score, mte) ----
    {
      const int st = hi0 - base;  // slot of the top cell
      unsigned uh = 0, vu = 0;
      // next row's top cell: does it move up?
      int hin = (r + 1 + w) >> 1;
      hin = hin > r + 1 ? r + 1 : hin;
      hin = hin > tlen - 1 ? tlen - 1 : hin;
      const bool up = hin == hi0 + 1 || hin == 0;
      const bool want_top = up || hi0 == tlen - 1 || hi0 == 0;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (want_top && (st >> 7) == k) uh = hi0 > 0 ? slot_half(U[k], st & 127) : slot_half(V[k], st & 127);
        if (!up && st > 0 && ((st - 1) >> 7) == k) vu = slot_half(V[k], (st - 1) & 127);
      }
      if (want_top) {
        if (r + T0 == 0) h_top = (int32_t)(uh >> 8) - 2 * sc.qe;
        else h_top = (hi0 > 0 ? h_under : h_top) + (int32_t)(uh >> 8) - sc.qe;
      }
      if (up || r + T0 == 0) {
        h_under = h_top;
      } else if (hi0 - 1 >= lo0) {
        h_under += (int32_t)(vu >> 8) - sc.qe;
      }
      if (hi0 == tlen - 1) {
        if (h_top > ez_mte) {
          ez_mte = h_top;
          ez_mte_q = r - hi;
        }
        if (r == nrow - 1) ez_score = h_top;
      }
    }
    if (r >= NSLOT - 1 && r < export_until) export_edge(r);
    prev_lo = lo;
    return true;
  };

  // ------------------------------------------------------------------------------------------
  // Lean rows [rb, re) of one block (rb >= 1): the same recurrence as the general row with the
  // rare cases taken out (row 0, captured carries, the sign-extension artefact -- the caller
  // routes those rows to slow_row) and everything that is constant over the segment hoisted:
  //   LOW16   the reference window starts at base+16 (lanes 0..7 of register 0 are out of it);
  //   SCALARH hi0 == tlen-1: the top cell's H is needed every row (mte / score) -> scalar path;
  //           otherwise the H path sum is accumulated inside the owning lane, reduced at the end;
  //   STEADY  pure band regime lo0 = (r-w+1)>>1, hi0 = (r+w)>>1, no boundary cell t = r, refresh
  //           range from register 0 to register KT: no per-register case analysis at all.
  // Lanes above the window top are NOT masked: they compute values nobody reads (the neighbour
  // dependency only runs upwards); the caller zeroes them when the window grows over them.
  // Lane predicates are VALU compares: the scalar unit is shared by the CU's four SIMDs and was
  // the bottleneck of the general row.
  // ------------------------------------------------------------------------------------------
  auto lean_rows = [&](auto low16_c, auto scalarh_c, auto steady_c, const int rb, const int re) {
    constexpr bool LOW16 = decltype(low16_c)::value;
    constexpr bool SCALARH = decltype(scalarh_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    constexpr int KT = NREG - 1;
    if (SCALARH) fold_h();
    if (qrow != rb) {  // (re)start the one-row-ahead query fetch at this row
      qaddr = (unsigned)(wofs + 2 * (qlen - 1 - rb + base + 32 + 2 * lane));
#pragma unroll
      for (int k = 0; k < NREG; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
    }
    qrow = re;
    if (STEADY && !SCALARH) hcnt += re - rb;  // every steady row takes one path step
    const unsigned vcar = base == 0 ? ((unsigned)sc.q_b << 24) : 0u;  // v carry into slot 0 (r > 0)
    const bool feed = has_left && base == 0;  // ... or the left stripe's edge values of the previous global row
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
      qaddr -= 2;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        qcur[k] = qnext[k];
        qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
      }
      // boundary cell t = r: y = 0, u = gap open (reference :122)
      if (!STEADY && off_hi + base >= r) {
        const int sr = r - base;
        const unsigned keep = (sr & 1) ? 0x0000ffffu : 0xffff0000u;
        const unsigned uval = ((unsigned)sc.q_b << 8) << ((sr & 1) * 16);
#pragma unroll
        for (int k = 0; k < NREG; ++k) {  // selects on every register (see slow_row)
          const bool mine = (sr >> 7) == k && lane == ((sr & 127) >> 1);
          U[k] = mine ? ((U[k] & keep) | uval) : U[k];
          Y[k] = mine ? (Y[k] & keep) : Y[k];
        }
      }
      unsigned xt1[NREG], vt1[NREG];
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        unsigned xs, vs;
        if (k == 0) {
          if (STEADY) {
            xs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X[0], 0x138, 0xf, 0xf, true);
            vs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)V[0], 0x138, 0xf, 0xf, true);
          } else if (feed) {
            const uint32_t fe = ring_in[(r + T0 - 1) & (SDF_RING - 1)];
            xs = (unsigned)__builtin_amdgcn_update_dpp((int)(fe << 16), (int)X[0], 0x138, 0xf, 0xf, false);
            vs = (unsigned)__builtin_amdgcn_update_dpp((int)(fe & 0xffff0000u), (int)V[0], 0x138, 0xf, 0xf, false);
          } else {
            xs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X[0], 0x138, 0xf, 0xf, true);
            vs = (unsigned)__builtin_amdgcn_update_dpp((int)vcar, (int)V[0], 0x138, 0xf, 0xf, false);
          }
        } else {
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
      // scores: refreshed slots are [ra, rbe)
      const int ra = lo0 - base;
      const int rbe = ra + ((hi0 - lo0) & ~15) + 16;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        const int b_ = rbe - 128 * k;
        if (STEADY) {
          unsigned z;
          SDF_FRESH(z, Tc[k], qcur[k])
          if (NREG == 1) {
            sel_lo_rng(S[0], z, (ra + 1) >> 1, (b_ + 1) >> 1, lane);
            sel_hi_rng(S[0], z, ra >> 1, b_ >> 1, lane);
          } else if (k == 0) {
            sel2_ge(S[0], z, (ra + 1) >> 1, ra >> 1, lane);
          } else if (k == KT) {
            sel2_lt(S[k], z, (b_ + 1) >> 1, b_ >> 1, lane);
          } else {
            S[k] = z;
          }
        } else if (b_ > 0) {
          unsigned z;
          SDF_FRESH(z, Tc[k], qcur[k])
          if (k == 0) {
            if (b_ >= 128) {
              sel2_ge(S[0], z, (ra + 1) >> 1, ra >> 1, lane);
            } else {
              sel_lo_rng(S[0], z, (ra + 1) >> 1, (b_ + 1) >> 1, lane);
              sel_hi_rng(S[0], z, ra >> 1, b_ >> 1, lane);
            }
          } else if (b_ >= 128) {
            S[k] = z;
          } else {
            sel2_lt(S[k], z, (b_ + 1) >> 1, b_ >> 1, lane);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        if (STEADY || off_hi >= 128 * k) {
          if (k == 0 && LOW16) {
            if (lane >= 8) SDF_CORE(0)
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
          if ((st >> 7) == k) uh = slot_half(U[k], st & 127);
          if (((st - 1) >> 7) == k) vu = slot_half(V[k], (st - 1) & 127);
        }
        h_top = h_under + (int32_t)(uh >> 8) - sc.qe;
        if (hi0 - 1 >= lo0) h_under += (int32_t)(vu >> 8) - sc.qe;
        if (h_top > ez_mte) {
          ez_mte = h_top;
          ez_mte_q = r - (hi0 | 15);
        }
        if (r == nrow - 1) ez_score = h_top;
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
          const int sh = ((sl & 1) << 4) + 8;
          unsigned val = 0u;
          if (STEADY) {
            const int slt = sl - 128 * KT;
            if (NREG > 1 && slt < 0) {
              if (up) val = U[KT > 0 ? KT - 1 : 0]; else val = V[KT > 0 ? KT - 1 : 0];
            } else {
              if (up) val = U[KT]; else val = V[KT];
            }
          } else {
#pragma unroll
            for (int k = 0; k < NREG; ++k)
              if ((sl >> 7) == k) {
                if (up) val = U[k]; else val = V[k];
              }
          }
          if (lane == ((sl & 127) >> 1)) hacc += (val >> sh) & 0xffu;
          if (!STEADY) ++hcnt;
        }
      }
      if (r >= NSLOT - 1 && r < export_until) export_edge(r);
    }
  };
  // U,V,X,Y of the cells t in [t_from, t_to] back to "never computed" (both bounds block aligned)
  auto zero_cells = [&](const int t_from, const int t_to) {
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int a_ = t_from - base - 128 * k, b_ = t_to - base - 128 * k;
      if (b_ >= 0 && a_ < 128) {
        const int la = a_ <= 0 ? 0 : a_ >> 1, lb = b_ >> 1;
        if ((unsigned)(lane - la) <= (unsigned)(lb - la)) {
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

  for (r0 = 0; r0 < nrow && !ez_zdropped; r0 += 16) {
    // ---- block start: re-base the window to the reference's band start of this row ----
    {
      Band b0;
      if (!band_of(r0, qlen, tlen, w, b0)) {
        ez_zdropped = 1;
        break;
      }
      carry_x = carry_v = 0u;
      if (b0.lo != base) {  // always +16: shift everything down by 8 lanes
        if (prev_lo == base) {  // the reference re-bases at this very row: its carry-in is the
          carry_x = slot_half(X[0], 15);  // (r-1) value of the cell just below the new window
          carry_v = slot_half(V[0], 15);
        }
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          const bool from_next = lane >= 56;
          unsigned a0, a1;
#define SDF_SHIFT8(A, INIT)                                                              \
  a0 = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_idx, (int)A[k]);                     \
  a1 = (k + 1 < NREG) ? (unsigned)__builtin_amdgcn_ds_bpermute(bperm_idx, (int)A[k + 1 < NREG ? k + 1 : k]) : (INIT); \
  A[k] = from_next ? a1 : a0;
          SDF_SHIFT8(U, 0u)
          SDF_SHIFT8(V, 0u)
          SDF_SHIFT8(X, 0u)
          SDF_SHIFT8(Y, 0u)
          SDF_SHIFT8(S, z_wild)
#undef SDF_SHIFT8
        }
        base = b0.lo;
        qrow = -1;  // the window moved: query addresses change
#pragma unroll
        for (int k = 0; k < NREG; ++k)
          Tc[k] = *reinterpret_cast<const uint32_t *>(Tb + (base - tt0) + 128 * k + 2 * lane);
        zero_low = false;
      }
    }
    // ---- systolic hand-shake with the neighbour stripes (whole blocks: 16 rows of slack either way) ----
    {
      const int g_end = T0 + (r0 + 15 < nrow - 1 ? r0 + 15 : nrow - 1);  // last global row of this block
      if (has_left) {
        if (base == 0) wait_ge(prog_prod + sb - 1, g_end - 1);  // the left stripe's edge of the rows before ours
      }
      if (has_right && r0 < export_until) wait_ge(prog_cons + sb, g_end - SDF_RING + 1);  // ring not overrun
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
        steady = lo0a + ((w - 1) & ~15) + 16 - base >= 128 * KT && (hi0a | 15) - base >= 128 * KT &&
                 hi0a - 1 - base >= 128 * KT - 128;
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
        bool special = !lean_ok || r + T0 == 0 || (r == r0 && (carry_x | carry_v) != 0u);
        if (rebase_row && !special) {
          // natural neighbour, but mind the sign-extension artefact of a negative carry
          const unsigned cvh = lo - base == 16 ? slot_half(V[0], 15) : carry_v;
          special = (cvh & 0x8000u) != 0u;
        }
        if (special) {
          if (dirty_hi > win_hi) zero_cells(win_hi + 1, dirty_hi);
          dirty_hi = win_hi;
          if (!slow_row(r)) {
            ez_zdropped = 1;
            break;
          }
          low16 = prev_lo - base == 16;
          if (has_right && track_h && r >= tlen - 1) {
            fold_h();
            hand_val[sb] = h_top;
            track_h = false;
          }
          ++r;
          continue;
        }
        if (rebase_row) {
          if (lo - base == 16) {
            drop_row = r;
            low16 = true;
          }
        } else if (low16 && !zero_low) {
          if (lane < 8) {
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
        }
        if (has_right && track_h && r == tlen - 1 && stop > r + 1) stop = r + 1;  // that row alone: its H is handed on
        const bool scalarh = track_h && hi0 == tlen - 1;
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
        if (has_right && track_h && stop > tlen - 1) {  // row NSLOT-1 done: the next stripe takes over the top cell
          fold_h();
          hand_val[sb] = h_top;
          track_h = false;
        }
        {  // the top register of the window now holds scratch values above the window
          const int top = base + 128 * (((win_hi - base) >> 7) + 1) - 1;
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
      if (drop_row >= 0 && lane < 8) {  // lanes that stopped shifting when their slots were dropped
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
          dir[((int64_t)rbk * NREG + k) * 64 + lane] = make_uint4(fa, fb, fx, fy);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NREG; ++k) Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    // publish: rows done (edge values and hand-off written before), edge rows consumed
    __threadfence_block();
    if (lane == 0) {
      if (has_right) prog_prod[sb] = T0 + r - 1;
      if (has_left) prog_cons[sb - 1] = (base == 0 && r < nrow) ? T0 + r - 2 : 0x7fffffff;
    }
  }
  __threadfence_block();
  if (lane == 0) {  // (a stripe that ends early must not hold its neighbours up)
    if (has_right) prog_prod[sb] = 0x7fffffff;
    if (has_left) prog_cons[sb - 1] = 0x7fffffff;
  }
  if (has_right) return;  // the last stripe owns the end of the target: score, mte

  fold_h();
  if (lane == 0) {
    sdf_result o;
    o.score = ez_score;
    o.max = 0;
    o.max_q = o.max_t = -1;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = ez_mte;
    o.mte_q = ez_mte_q;
    o.zdropped = ez_zdropped;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[tk.out_idx] = o;
  }
}

#undef tt0
#undef we0

template __global__ void extz2_stripe_kernel<1>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                sdf_result *);
template __global__ void extz2_stripe_kernel<2>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                sdf_result *);
template __global__ void extz2_stripe_kernel<4>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                sdf_result *);

// dynamic LDS of a launch with `nstripe` wavefronts per workgroup
size_t stripe_lds_bytes(int qlen, int nstripe, int nreg) {
  const size_t nslot = 128 * (size_t)nreg;
  return 256 + 16 * SDF_RING * 4 + ((2 * ((size_t)qlen + nslot + 36) + 15) & ~(size_t)15) + (size_t)nstripe * 2 * (2 * nslot + 32);
}

// flag bytes of one stripe's region (sized for a full stripe)
size_t stripe_dir_bytes(int qlen, int nreg) {
  return (size_t)((qlen + 128 * nreg - 1 + 15) / 16) * (size_t)nreg * 1024;
}

}  // namespace sdf
