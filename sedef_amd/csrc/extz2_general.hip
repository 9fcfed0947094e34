// General extz2 DP kernel for gfx950: any band, any length that fits LDS, every ksw_extz_t field.
//
// What it computes: the anti-diagonal difference-form affine-gap DP of the reference kernel
// (reference: extern/ksw2_extz2_sse.cc:23-298) with bit-identical results, including the
// cells the reference computes outside the logical band because it works in 16-cell blocks
// (:115) and the scores it leaves stale there (:124-138).  To make that exact by construction
// the per-target-position state lives in LDS in the same order the reference keeps it
// (u|v|x|y|s|target|reversed query, zero-initialised), so every read "past the end" of one
// array lands where it lands in the reference.
//
// Mapping: one workgroup (64 or 256 threads) per DP task.  An anti-diagonal's widened band is
// cut into chunks of 4*BS cells; a thread owns 4 consecutive cells (one dword per state array).
// Chunks are visited from high t to low t so that the (r-1, t-1) neighbour of a chunk's first
// cell is still the previous anti-diagonal's value.  Direction bytes (1 B/cell, row stride
// n_col*16 as in the reference) stream to HBM as one coalesced dword per thread.
//
// This is the correctness-first kernel and the fallback for shapes the register-resident
// kernel (extz2_wave.hip) does not take.
#include <hip/hip_runtime.h>

#include "sdf_internal.h"

namespace sdf {

struct BestCell {
  int32_t H, r, key, t;
};

// a beats b: larger H; then earlier anti-diagonal; then the reference's in-row scan order
__device__ __forceinline__ bool beats(const BestCell &a, const BestCell &b) {
  if (a.H != b.H) return a.H > b.H;
  if (a.r != b.r) return a.r < b.r;
  return a.key < b.key;
}

__device__ __forceinline__ BestCell wave_best(BestCell c) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    BestCell o;
    o.H = __shfl_xor(c.H, off);
    o.r = __shfl_xor(c.r, off);
    o.key = __shfl_xor(c.key, off);
    o.t = __shfl_xor(c.t, off);
    if (beats(o, c)) c = o;
  }
  return c;
}

template <int BS>
__device__ __forceinline__ BestCell block_best(BestCell c, BestCell *red) {
  c = wave_best(c);
  if (BS > 64) {
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wv] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < BS / 64; ++k)
        if (beats(red[k], c)) c = red[k];
    }
    __syncthreads();
  }
  return c;  // valid in thread 0
}

__device__ __forceinline__ uint32_t packed_code(const uint32_t *codes, const uint32_t *nmask, int k,
                                                uint32_t wild) {
  uint32_t c = (codes[k >> 4] >> ((k & 15) * 2)) & 3u;
  uint32_t n = (nmask[k >> 5] >> (k & 31)) & 1u;
  return n ? wild : c;
}

// GLOBAL: the arena and H[] of tasks too long for LDS (> ~14k) live in an HBM scratch slab per workgroup;
// same code, same barriers (they order the workgroup's global accesses as well).
// The recurrence of tasks with left-aligned gaps runs on the packed 16-bit ALU, two cells per instruction as in
// extz2_wave.hip.  PLAIN: only CIGAR / score / mte are wanted, zdrop < 0, left-aligned gaps (what SEDEF asks for):
// the exact H is followed along the band's upper edge only (thread 0, O(1) per row) instead of being updated for
// every cell of every row.  The memory layout, the order of the LDS accesses and every artefact of the reference
// stay as in the full version.
template <int BS, bool GLOBAL, bool PLAIN>
__global__ __launch_bounds__(BS) void extz2_general_kernel(
    const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
    const uint32_t *__restrict__ pool, ScoreK sc, uint8_t *__restrict__ dirbase,
    sdf_result *__restrict__ res, uint8_t *__restrict__ gscratch, size_t gstride) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  uint8_t *lds = GLOBAL ? gscratch + (size_t)blockIdx.x * gstride : lds_raw;
  const PlanTask tk = plan[order[blockIdx.x]];
  const int tid = threadIdx.x;
  const int qlen = tk.qlen, tlen = tk.tlen, w = tk.w;
  const int T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
  uint8_t *U = lds, *V = U + T16, *X = V + T16, *Y = X + T16, *S = Y + T16;
  uint8_t *SF = S + T16, *QR = SF + T16;
  const int arena = 6 * T16 + Q16 + 16;
  int32_t *H = reinterpret_cast<int32_t *>(lds + arena);
  BestCell *red = GLOBAL ? reinterpret_cast<BestCell *>(lds_raw) : reinterpret_cast<BestCell *>(H + T16);
  int *stop_flag = reinterpret_cast<int *>(red + 16);  // one slot per wavefront of the largest workgroup (1024)

  for (int k = tid * 4; k < arena; k += BS * 4) *reinterpret_cast<uint32_t *>(lds + k) = 0u;
  if (!PLAIN)
    for (int k = tid; k < T16; k += BS) H[k] = SDF_NEG_INF;
  if (tid == 0) *stop_flag = 0;
  __syncthreads();
  {
    const uint32_t *tw = pool + tk.t_word, *tn = tw + (tlen + 15) / 16;
    const uint32_t *qw = pool + tk.q_word, *qn = qw + (qlen + 15) / 16;
    for (int k = tid; k < tlen; k += BS) SF[k] = (uint8_t)packed_code(tw, tn, k, sc.wild);
    for (int k = tid; k < qlen; k += BS) QR[k] = (uint8_t)packed_code(qw, qn, qlen - 1 - k, sc.wild);
  }
  __syncthreads();

  const int flag = tk.flag;
  const bool with_dir = !(flag & SDF_FLAG_SCORE_ONLY);
  const bool right = (flag & SDF_FLAG_RIGHT) != 0;
  // KSW_EZ_GENERIC_SC: scores from the whole matrix; KSW_EZ_APPROX_MAX: no H[] at all, one H value followed from cell to
  // cell (reference :268-283), with KSW_EZ_APPROX_DROP the z-drop test on it
  const bool generic_sc = !PLAIN && (flag & SDF_FLAG_GENERIC_SC) != 0;
  const bool approx = !PLAIN && (flag & SDF_FLAG_APPROX_MAX) != 0;
  const bool zd_mode = tk.zdrop >= 0 && !approx;
  int32_t H0 = 0;
  int last_H0_t = 0;
  const int64_t stride = tk.ncol16;
  uint8_t *dir = dirbase + tk.dir_off;
  const int nrow = qlen + tlen - 1;

  // ksw_extz_t state, meaningful in thread 0
  int32_t ez_max = 0, ez_max_t = -1, ez_max_q = -1;
  int32_t ez_mqe = SDF_NEG_INF, ez_mqe_t = -1, ez_mte = SDF_NEG_INF, ez_mte_q = -1;
  int32_t ez_score = SDF_NEG_INF, ez_zdropped = 0;
  BestCell best = {0, -1, 0, -1};  // running arg-max over all cells (zdrop < 0 mode)
  int prev_lo = -1, prev_hi = -1;
  int32_t h_top = 0, h_under = 0;  // PLAIN: H of the top cell / of the cell the next top cell reads (thread 0)
  // PLAIN: constants of the << 8 difference domain
  const unsigned qv2 = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned capv2 = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned qe2v = ((unsigned)sc.qe2_b << 8) * 0x00010001u;
  unsigned one2 = 0x00010001u;
  SDF_OPQ(one2);

  for (int r = 0; r < nrow; ++r) {
    Band b;
    if (!band_of(r, qlen, tlen, w, b)) {
      ez_zdropped = 1;
      break;
    }
    // ---- carry-in of cell `lo`, boundary writes, score refresh ----
    int cx, cv;
    if (b.lo > 0) {
      if (b.lo - 1 >= prev_lo && b.lo - 1 <= prev_hi) {
        cx = (int8_t)X[b.lo - 1];
        cv = (int8_t)V[b.lo - 1];
      } else {
        cx = cv = 0;
      }
    } else {
      cx = 0;
      cv = r ? sc.q : 0;
    }
    int32_t h_diag_prev = 0;
    if (tid == 0) {
      if (!PLAIN && !approx) h_diag_prev = b.hi0 > 0 ? H[b.hi0 - 1] : H[b.hi0];
      if (b.hi >= r) {
        Y[r] = 0;
        U[r] = r ? sc.q_b : 0;
      }
    }
    if (generic_sc) {  // exactly [st0, en0], no 16-cell rounding (reference :139-141)
      const uint8_t *qrow = QR + (qlen - 1 - r);
      for (int t = b.lo0 + tid; t <= b.hi0; t += BS) S[t] = (uint8_t)sc.mat[SF[t] * 5 + qrow[t]];
    } else {
      const int top = b.lo0 + ((b.hi0 - b.lo0) / 16 + 1) * 16;
      const uint8_t *qrow = QR + (qlen - 1 - r);
      for (int t = b.lo0 + tid; t < top; t += BS) {
        const uint8_t a = SF[t], c = qrow[t];
        uint8_t s = a == c ? sc.sc_match : sc.sc_mis;
        if (a == sc.wild || c == sc.wild) s = 0;
        S[t] = s;  // t >= T16 spills into SF[0..14]; those cells are below lo0 by then
      }
    }
    __syncthreads();

    // ---- recurrence over the widened band, 4 cells per thread, chunks from high t down ----
    const int ncell = b.hi - b.lo + 1;
    const int nchunk = (ncell + 4 * BS - 1) / (4 * BS);
    for (int c = nchunk - 1; c >= 0; --c) {
      const int t = b.lo + c * 4 * BS + tid * 4;
      const bool act = t <= b.hi;
      uint32_t u4 = 0, v4 = 0, x4 = 0, y4 = 0, s4 = 0;
      uint32_t xl = 0, vl = 0;
      if (act) {
        u4 = *reinterpret_cast<const uint32_t *>(U + t);
        v4 = *reinterpret_cast<const uint32_t *>(V + t);
        x4 = *reinterpret_cast<const uint32_t *>(X + t);
        y4 = *reinterpret_cast<const uint32_t *>(Y + t);
        s4 = *reinterpret_cast<const uint32_t *>(S + t);
        if (t == b.lo) {
          xl = (uint32_t)cx & 0xffu;
          vl = (uint32_t)cv & 0xffu;
        } else {
          xl = X[t - 1];
          vl = V[t - 1];
        }
      }
      __syncthreads();
      if ((PLAIN || !right) && act) {  // left-aligned gaps (every SEDEF call): the packed 16-bit recurrence
        // bytes -> value << 8 in 16-bit halves: cells (0,1) and (2,3) of the thread
        const uint32_t smx = (t == b.lo && cx < 0) ? 0xff00u : 0u, smv = (t == b.lo && cv < 0) ? 0xff00u : 0u;
        uint32_t un[2], vn[2], xn[2], yn[2], dd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t sel = h ? 0x030c020cu : 0x010c000cu;
          const uint32_t uo = __builtin_amdgcn_perm(0u, u4, sel), yo = __builtin_amdgcn_perm(0u, y4, sel);
          const uint32_t so = __builtin_amdgcn_perm(0u, s4, sel);
          // (r-1, t-1) neighbours: (carry, cell 0) for the low pair, (cell 1, cell 2) for the high pair; a negative
          // carry byte is sign-extended over cells 1..3 of the first block (reference :145-146)
          uint32_t xt1 = h ? __builtin_amdgcn_perm(0u, x4, 0x020c010cu) : __builtin_amdgcn_perm(x4, xl, 0x040c000cu);
          uint32_t vt1 = h ? __builtin_amdgcn_perm(0u, v4, 0x020c010cu) : __builtin_amdgcn_perm(v4, vl, 0x040c000cu);
          xt1 |= h ? smx * 0x00010001u : smx << 16;
          vt1 |= h ? smv * 0x00010001u : smv << 16;
          const uint32_t z0 = pk_add(so, qe2v);
          const uint32_t a_ = pk_add(xt1, vt1), bb_ = pk_add(yo, uo);
          const uint32_t z1 = pk_maxi(z0, a_);
          const uint32_t fa_ = pk_sub(z1, z0);  // != 0 <=> a > z (signed)
          const uint32_t zb_ = pk_maxi(z1, bb_);
          const uint32_t fb_ = pk_sub(zb_, z1);  // != 0 <=> b > max(z, a)
          const uint32_t z3 = pk_minu(pk_maxu(z1, bb_), capv2);
          un[h] = pk_sub(z3, vt1);
          vn[h] = pk_sub(z3, uo);
          const uint32_t zq = pk_sub(z3, qv2);
          xn[h] = pk_maxi(pk_sub(a_, zq), 0u);
          yn[h] = pk_maxi(pk_sub(bb_, zq), 0u);
          // direction byte: 2 if b won, else 1 if a won; | 0x08 if x > 0; | 0x10 if y > 0
          const uint32_t fa1 = pk_nonzero(fa_), fb1 = pk_nonzero(fb_);
          uint32_t d = pk_maxu(fa1, pk_add(fb1, fb1));
          d = pk_mad(pk_nonzero(xn[h]), 0x00080008u, d);
          d = pk_mad(pk_nonzero(yn[h]), 0x00100010u, d);
          dd[h] = d;
        }
        // value << 8 halves -> bytes (bytes 1, 3 of each register); direction bytes are bytes 0, 2
        *reinterpret_cast<uint32_t *>(U + t) = __builtin_amdgcn_perm(un[1], un[0], 0x07050301u);
        *reinterpret_cast<uint32_t *>(V + t) = __builtin_amdgcn_perm(vn[1], vn[0], 0x07050301u);
        *reinterpret_cast<uint32_t *>(X + t) = __builtin_amdgcn_perm(xn[1], xn[0], 0x07050301u);
        *reinterpret_cast<uint32_t *>(Y + t) = __builtin_amdgcn_perm(yn[1], yn[0], 0x07050301u);
        if (with_dir)
          *reinterpret_cast<uint32_t *>(dir + (int64_t)r * stride + (t - b.lo)) =
              __builtin_amdgcn_perm(dd[1], dd[0], 0x06040200u);
      }
      if (!PLAIN && right && act) {  // right-aligned gaps: scalar byte arithmetic
        // a negative carry byte is sign-extended over lanes 1..3 of the first block
        const uint32_t smx = (t == b.lo && cx < 0) ? 0xffu : 0u;
        const uint32_t smv = (t == b.lo && cv < 0) ? 0xffu : 0u;
        uint32_t un = 0, vn = 0, xn = 0, yn = 0, d4 = 0;
        uint32_t xprev = xl, vprev = vl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t xo = (x4 >> (8 * k)) & 0xffu, vo = (v4 >> (8 * k)) & 0xffu;
          const uint32_t uo = (u4 >> (8 * k)) & 0xffu, yo = (y4 >> (8 * k)) & 0xffu;
          const uint32_t so = (s4 >> (8 * k)) & 0xffu;
          uint32_t xt1 = xprev, vt1 = vprev;
          if (k >= 1) {
            xt1 |= smx;
            vt1 |= smv;
          }
          xprev = xo;
          vprev = vo;
          uint32_t z = (so + sc.qe2_b) & 0xffu;
          uint32_t a = (xt1 + vt1) & 0xffu;
          uint32_t bb = (yo + uo) & 0xffu;
          uint32_t d;
          if (!right) {
            d = (int8_t)a > (int8_t)z ? 1u : 0u;
            if ((int8_t)a > (int8_t)z) z = a;
            if ((int8_t)bb > (int8_t)z) d = 2u;
          } else {
            d = (int8_t)z > (int8_t)a ? 0u : 1u;
            if ((int8_t)a > (int8_t)z) z = a;
            if (!((int8_t)z > (int8_t)bb)) d = 2u;
          }
          if (bb > z) z = bb;
          if (z > sc.cap_b) z = sc.cap_b;
          const uint32_t unew = (z - vt1) & 0xffu;
          const uint32_t vnew = (z - uo) & 0xffu;
          z = (z - sc.q_b) & 0xffu;
          a = (a - z) & 0xffu;
          bb = (bb - z) & 0xffu;
          uint32_t xnew, ynew;
          if (!right) {
            xnew = (int8_t)a > 0 ? a : 0u;
            ynew = (int8_t)bb > 0 ? bb : 0u;
            if ((int8_t)a > 0) d |= 0x08u;
            if ((int8_t)bb > 0) d |= 0x10u;
          } else {
            xnew = (int8_t)a < 0 ? 0u : a;
            ynew = (int8_t)bb < 0 ? 0u : bb;
            if (!((int8_t)a < 0)) d |= 0x08u;
            if (!((int8_t)bb < 0)) d |= 0x10u;
          }
          un |= unew << (8 * k);
          vn |= vnew << (8 * k);
          xn |= xnew << (8 * k);
          yn |= ynew << (8 * k);
          d4 |= d << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(U + t) = un;
        *reinterpret_cast<uint32_t *>(V + t) = vn;
        *reinterpret_cast<uint32_t *>(X + t) = xn;
        *reinterpret_cast<uint32_t *>(Y + t) = yn;
        if (with_dir) *reinterpret_cast<uint32_t *>(dir + (int64_t)r * stride + (t - b.lo)) = d4;
      }
    }
    __syncthreads();

    // ---- exact H[] and the row arg-max (reference :222-258) ----
    BestCell rowbest = {SDF_NEG_INF, r, 0x7fffffff, -1};
    if (PLAIN) {
      // exact H of the top cell and of the cell under the band edge only (see slow_row in extz2_wave.hip)
      if (tid == 0) {
        int hin = (r + 1 + w) >> 1;  // next row's top cell: does it move up?
        hin = hin > r + 1 ? r + 1 : hin;
        hin = hin > tlen - 1 ? tlen - 1 : hin;
        const bool up = hin == b.hi0 + 1 || hin == 0;
        const bool want_top = up || b.hi0 == tlen - 1 || b.hi0 == 0;
        if (want_top) {
          const int32_t uh = b.hi0 > 0 ? (int32_t)U[b.hi0] : (int32_t)V[b.hi0];
          if (r == 0) h_top = uh - 2 * sc.qe;
          else h_top = (b.hi0 > 0 ? h_under : h_top) + uh - sc.qe;
        }
        if (up || r == 0) h_under = h_top;
        else if (b.hi0 - 1 >= b.lo0) h_under += (int32_t)V[b.hi0 - 1] - sc.qe;
        if (b.hi0 == tlen - 1) {
          if (h_top > ez_mte) {
            ez_mte = h_top;
            ez_mte_q = r - b.hi;
          }
          if (r == nrow - 1) ez_score = h_top;
        }
      }
    } else if (approx) {
      if (tid == 0) {
        const bool in0 = last_H0_t >= b.lo0 && last_H0_t <= b.hi0, in1 = last_H0_t + 1 >= b.lo0 && last_H0_t + 1 <= b.hi0;
        if (r == 0) {
          H0 = (int32_t)V[0] - 2 * sc.qe;
          last_H0_t = 0;
        } else if (in0 && in1) {
          const int32_t d0 = (int32_t)V[last_H0_t] - sc.qe, d1 = (int32_t)U[last_H0_t + 1] - sc.qe;
          if (d0 > d1) {
            H0 += d0;
          } else {
            H0 += d1;
            ++last_H0_t;
          }
        } else if (in0) {
          H0 += (int32_t)V[last_H0_t] - sc.qe;
        } else {
          ++last_H0_t;
          H0 += (int32_t)U[last_H0_t] - sc.qe;
        }
      }
    } else if (r > 0) {
      const int vec_end = b.lo0 + (b.hi0 - b.lo0) / 4 * 4;
      for (int t = b.lo0 + tid; t < b.hi0; t += BS) {
        const int32_t h = H[t] + (int32_t)V[t] - sc.qe;
        H[t] = h;
        BestCell cnd = {h, r, t < vec_end ? 1 + (((t - b.lo0) & 3) << 20) + t : 1 + (4 << 20) + t, t};
        if (beats(cnd, rowbest)) rowbest = cnd;
      }
      if (tid == 0) {
        const int32_t h = h_diag_prev + (b.hi0 > 0 ? (int32_t)U[b.hi0] : (int32_t)V[b.hi0]) - sc.qe;
        H[b.hi0] = h;
        BestCell cnd = {h, r, 0, b.hi0};
        if (beats(cnd, rowbest)) rowbest = cnd;
      }
    } else if (tid == 0) {
      const int32_t h = (int32_t)V[0] - 2 * sc.qe;
      H[0] = h;
      rowbest = BestCell{h, 0, 0, 0};
    }
    if (!zd_mode) {
      if (rowbest.t >= 0 && beats(rowbest, best)) best = rowbest;
    }
    __syncthreads();

    // ---- ksw_extz_t bookkeeping (reference :259-267) ----
    if (zd_mode) rowbest = block_best<BS>(rowbest, red);
    if (!PLAIN && approx) {  // (mte / mqe are not produced in this mode; max only through the z-drop test, r > 0)
      if (tid == 0) {
        bool stop = false;
        if (r > 0 && (flag & SDF_FLAG_APPROX_DROP)) {  // ksw_apply_zdrop on the followed value
          const int tt = last_H0_t;
          if (H0 > ez_max) {
            ez_max = H0;
            ez_max_t = tt;
            ez_max_q = r - tt;
          } else if (tt >= ez_max_t && r - tt >= ez_max_q) {
            const int tl = tt - ez_max_t, ql = (r - tt) - ez_max_q;
            const int l = tl > ql ? tl - ql : ql - tl;
            if (tk.zdrop >= 0 && ez_max - H0 > tk.zdrop + l * sc.e) {
              ez_zdropped = 1;
              stop = true;
            }
          }
          if (stop) *stop_flag = 1;
        }
        if (!stop && r == nrow - 1 && b.hi0 == tlen - 1) ez_score = H0;
      }
      if (flag & SDF_FLAG_APPROX_DROP) {
        __syncthreads();
        if (*stop_flag) break;
      }
    } else if (!PLAIN && tid == 0) {
      if (b.hi0 == tlen - 1 && H[b.hi0] > ez_mte) {
        ez_mte = H[b.hi0];
        ez_mte_q = r - b.hi;
      }
      if (r - b.lo0 == qlen - 1 && H[b.lo0] > ez_mqe) {
        ez_mqe = H[b.lo0];
        ez_mqe_t = b.lo0;
      }
      bool stop = false;
      if (zd_mode) {  // ksw_apply_zdrop (reference: extern/ksw2.h:161-177)
        const int32_t hh = rowbest.H, tt = rowbest.t;
        if (hh > ez_max) {
          ez_max = hh;
          ez_max_t = tt;
          ez_max_q = r - tt;
        } else if (tt >= ez_max_t && r - tt >= ez_max_q) {
          const int tl = tt - ez_max_t, ql = (r - tt) - ez_max_q;
          const int l = tl > ql ? tl - ql : ql - tl;
          if (ez_max - hh > tk.zdrop + l * sc.e) {
            ez_zdropped = 1;
            stop = true;
          }
        }
        if (stop) *stop_flag = 1;
      }
      if (!stop && r == nrow - 1 && b.hi0 == tlen - 1) ez_score = H[tlen - 1];
    }
    if (zd_mode) {
      __syncthreads();
      if (*stop_flag) break;
    }
    prev_lo = b.lo;
    prev_hi = b.hi;
  }

  if (!PLAIN && !zd_mode) {
    best = block_best<BS>(best, red);
    if (tid == 0 && best.r >= 0) {
      ez_max = best.H;
      ez_max_t = best.t;
      ez_max_q = best.r - best.t;
    }
  }
  if (tid == 0) {
    sdf_result o;
    o.score = ez_score;
    o.max = ez_max;
    o.max_q = ez_max_q;
    o.max_t = ez_max_t;
    o.mqe = ez_mqe;
    o.mqe_t = ez_mqe_t;
    o.mte = ez_mte;
    o.mte_q = ez_mte_q;
    o.zdropped = ez_zdropped;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[tk.out_idx] = o;
  }
}

template __global__ void extz2_general_kernel<64, false, false>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<256, false, false>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<1024, false, false>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<1024, true, false>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<256, true, false>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<1024, false, true>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<1024, true, true>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);
template __global__ void extz2_general_kernel<256, false, true>(const PlanTask *, const int32_t *, const uint32_t *,
                                                                ScoreK, uint8_t *, sdf_result *, uint8_t *, size_t);

size_t general_lds_bytes(int qlen, int tlen) {
  const size_t T16 = (size_t)(tlen + 15) / 16 * 16, Q16 = (size_t)(qlen + 15) / 16 * 16;
  return 6 * T16 + Q16 + 16 + 4 * T16 + 16 * sizeof(BestCell) + 16;
}

}  // namespace sdf
