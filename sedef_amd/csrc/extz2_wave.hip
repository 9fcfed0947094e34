// Register-resident extz2 DP kernel for gfx950: one wavefront per DP task.
//
// Same results as extz2_general.hip (and therefore as the reference kernel,
// extern/ksw2_extz2_sse.cc:23-298) for the fields SEDEF consumes -- CIGAR, score, mte -- but the
// anti-diagonal state never leaves the register file:
//
//   * a lane owns two adjacent cells (slots 2l, 2l+1) of every 128-slot register; each state
//     byte of the reference (u, v, x, y, and the possibly stale score s) is kept as value<<8 in a
//     16-bit half, so v_pk_{add,sub,max,min}_{u,i}16 reproduce the reference's wrap-around int8
//     arithmetic (signed and unsigned) exactly, two cells per instruction;
//   * slot k of the window is target position base+k, where base = the reference's block-rounded
//     band start at the first row of each 16-row block, so the window only re-bases (by 16 slots
//     = 8 lanes, via ds_bpermute) at block boundaries;
//   * the (r-1, t-1) neighbour comes from a wavefront DPP shift (wave_shr:1) + v_alignbit;
//   * the reversed query and the target sit in LDS as 16-bit codes (unpacked once from the
//     2-bit/N-mask pool with coalesced dword loads); one aligned ds_read_b32 per register per row
//     yields the two query codes of a lane;
//   * the four direction flags of a cell are shifted into four 32-bit accumulators (16 rows x
//     2 cells each) and leave for HBM as one 16-byte store per lane per 16 rows: 0.5 B/cell,
//     1 KiB per wave-instruction, fully coalesced.
//
// Exactness in banded mode: the reference computes whole 16-cell blocks, i.e. also cells outside
// the logical band, from persistent per-position state and from scores refreshed only in
// [st0, st0+16*n) (:115,:124-138).  Those cells feed real cells at the band edges and the
// traceback may walk through them, so they are computed here too: lanes are enabled for exactly
// the reference's widened range and the score register keeps its old value outside the refreshed
// range.  The exact H values the reference derives for score / mte are followed along the one
// path of cells they depend on (the cell under the band's upper edge), O(1) per row.
//
// Not produced here (general kernel instead): max/max_q/max_t, mqe/mqe_t, z-drop, right-aligned
// gaps, extension-only traceback.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "sdf_internal.h"

namespace sdf {

// lane mask with bits [lo, hi) set (0 <= lo, hi; clamped to 64)
__device__ __forceinline__ unsigned long long lane_range(int lo, int hi) {
  lo = lo < 0 ? 0 : lo;
  hi = hi > 64 ? 64 : hi;
  if (hi <= lo) return 0ull;
  const unsigned long long top = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
  return top & ~((1ull << lo) - 1ull);
}

// dst half <- src half where the lane's bit in `mask` is set (SDWA keeps the other half)
__device__ __forceinline__ void sel_lo16(unsigned &dst, unsigned src, unsigned long long mask) {
  asm volatile(
      "s_mov_b64 vcc, %2\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\ts_nop 0"
      : "+v"(dst)
      : "v"(src), "s"(mask)
      : "vcc");
}
__device__ __forceinline__ void sel_hi16(unsigned &dst, unsigned src, unsigned long long mask) {
  asm volatile(
      "s_mov_b64 vcc, %2\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1\n\ts_nop 0"
      : "+v"(dst)
      : "v"(src), "s"(mask)
      : "vcc");
}

// value of slot `s` (0..127) of a packed register, as its 16-bit half
__device__ __forceinline__ unsigned slot_half(unsigned reg, int s) {
  const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)reg, s >> 1);
  return (s & 1) ? (w >> 16) : (w & 0xffffu);
}

// S half <- z half in the lanes where `thr <= lane` (GE) or `lane < thr` (LT); the compare writes
// VCC and the SDWA select consumes it (no wait state needed between them on gfx9).
__device__ __forceinline__ void sel_lo_ge(unsigned &dst, unsigned src, int thr, int lane) {
  asm volatile(
      "v_cmp_le_i32 vcc, %2, %3\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\ts_nop 0"
      : "+v"(dst) : "v"(src), "s"(thr), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel_hi_ge(unsigned &dst, unsigned src, int thr, int lane) {
  asm volatile(
      "v_cmp_le_i32 vcc, %2, %3\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1\n\ts_nop 0"
      : "+v"(dst) : "v"(src), "s"(thr), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel_lo_lt(unsigned &dst, unsigned src, int thr, int lane) {
  asm volatile(
      "v_cmp_gt_i32 vcc, %2, %3\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\ts_nop 0"
      : "+v"(dst) : "v"(src), "s"(thr), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel_hi_lt(unsigned &dst, unsigned src, int thr, int lane) {
  asm volatile(
      "v_cmp_gt_i32 vcc, %2, %3\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1\n\ts_nop 0"
      : "+v"(dst) : "v"(src), "s"(thr), "v"(lane) : "vcc");
}

// two-sided: lanes in [lo, hi)
__device__ __forceinline__ void sel_lo_rng(unsigned &dst, unsigned src, int lo, int hi, int lane) {
  unsigned t;
  asm volatile(
      "v_subrev_u32 %1, %3, %5\n\t"
      "v_cmp_gt_u32 vcc, %4, %1\n\t"
      "v_cndmask_b32_sdwa %0, %0, %2, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\ts_nop 0"
      : "+v"(dst), "=&v"(t) : "v"(src), "s"(lo), "s"(hi > lo ? hi - lo : 0), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel_hi_rng(unsigned &dst, unsigned src, int lo, int hi, int lane) {
  unsigned t;
  asm volatile(
      "v_subrev_u32 %1, %3, %5\n\t"
      "v_cmp_gt_u32 vcc, %4, %1\n\t"
      "v_cndmask_b32_sdwa %0, %0, %2, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1\n\ts_nop 0"
      : "+v"(dst), "=&v"(t) : "v"(src), "s"(lo), "s"(hi > lo ? hi - lo : 0), "v"(lane) : "vcc");
}

// (the same with the range as start + length: the caller has the length in a register already)
__device__ __forceinline__ void sel_lo_len(unsigned &dst, unsigned src, int lo, int len, int lane) {
  unsigned t;
  asm volatile(
      "v_subrev_u32 %1, %3, %5\n\t"
      "v_cmp_gt_u32 vcc, %4, %1\n\t"
      "v_cndmask_b32_sdwa %0, %0, %2, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\ts_nop 0"
      : "+v"(dst), "=&v"(t) : "v"(src), "s"(lo), "s"(len), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel_hi_len(unsigned &dst, unsigned src, int lo, int len, int lane) {
  unsigned t;
  asm volatile(
      "v_subrev_u32 %1, %3, %5\n\t"
      "v_cmp_gt_u32 vcc, %4, %1\n\t"
      "v_cndmask_b32_sdwa %0, %0, %2, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1\n\ts_nop 0"
      : "+v"(dst), "=&v"(t) : "v"(src), "s"(lo), "s"(len), "v"(lane) : "vcc");
}

// (the range [0, bound): one signed compare)
__device__ __forceinline__ void sel_lo_below(unsigned &dst, unsigned src, int bound, int lane) {
  asm volatile(
      "v_cmp_gt_i32 vcc, %2, %3\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\ts_nop 0"
      : "+v"(dst) : "v"(src), "s"(bound), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel_hi_below(unsigned &dst, unsigned src, int bound, int lane) {
  asm volatile(
      "v_cmp_gt_i32 vcc, %2, %3\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1\n\ts_nop 0"
      : "+v"(dst) : "v"(src), "s"(bound), "v"(lane) : "vcc");
}

// both halves of one register: lo half where thr_lo <= lane, hi half where thr_hi <= lane
__device__ __forceinline__ void sel2_ge(unsigned &dst, unsigned src, int thr_lo, int thr_hi, int lane) {
  asm volatile(
      "v_cmp_le_i32 vcc, %2, %4\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\t"
      "v_cmp_le_i32 vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1"
      : "+v"(dst) : "v"(src), "s"(thr_lo), "s"(thr_hi), "v"(lane) : "vcc");
}
__device__ __forceinline__ void sel2_lt(unsigned &dst, unsigned src, int thr_lo, int thr_hi, int lane) {
  asm volatile(
      "v_cmp_gt_i32 vcc, %2, %4\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 "
      "src1_sel:WORD_0\n\t"
      "v_cmp_gt_i32 vcc, %3, %4\n\t"
      "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 "
      "src1_sel:WORD_1"
      : "+v"(dst) : "v"(src), "s"(thr_lo), "s"(thr_hi), "v"(lane) : "vcc");
}

__device__ __forceinline__ uint32_t pool_code16(const uint32_t *codes, const uint32_t *nmask, int k,
                                                uint32_t wild) {
  const uint32_t c = (codes[k >> 4] >> ((k & 15) * 2)) & 3u;
  const uint32_t n = (nmask[k >> 5] >> (k & 31)) & 1u;
  return n ? (0xff00u | wild) : c;
}

// ---- fresh scores as byte permutes (end of round 6; extz2_pair.hip: SDF_PFRESH has the story) ----------------------------
// The kernels whose lanes hold TWO ADJACENT target positions per register (wave, stripe, banded stripe): a table of four score
// bytes per position -- against query base 0..3; an N in the target: the wildcard's score four times --, the row's two query
// bases as a selector (byte 1 = base of the even position: a byte of the first table; byte 3 = 4 + base of the odd one: a byte
// of the second; bytes 0, 2 = 0x0c: zero; an N: 0xff, patched afterwards where the sequences hold any N).
// code: what pool_code16 returns (0..3, N: 0xff00 | wild)
__device__ __forceinline__ unsigned score_table(const unsigned code, const unsigned mis4, const unsigned delta, const unsigned wild4) {
  return (code & 0xff00u) ? wild4 : mis4 ^ (delta << (8u * code));
}
// an entry of a query window of byte pairs (W[i] = bases of window positions i, i + 1) in selector form
__device__ __forceinline__ uint16_t qsel_pair(const uint32_t v0, const uint32_t v1) {
  return (uint16_t)(((v0 & 0xff00u) ? 0xffu : v0) | (((v1 & 0xff00u) ? 0xffu : v1 + 4u) << 8));
}
// the two selector bytes of a window entry -> the permute's selector 0x0c, s0, 0x0c, s1
__device__ __forceinline__ unsigned qsel_spread(const unsigned w16) {
  return __builtin_amdgcn_perm(0x0c0c0c0cu, w16, 0x01040004u);
}
#define SDF_SCORE2(z, k, qs, WITH_N)                                    \
  {                                                                     \
    z = __builtin_amdgcn_perm(TB[k], TA[k], (qs));                      \
    if (WITH_N) {                                                       \
      unsigned nn_ = pk_ashr15(z);                                      \
      SDF_OPQ(nn_);                                                     \
      z = (z_wild & nn_) | (z & ~nn_);                                  \
    }                                                                   \
  }

// One anti-diagonal step of the recurrence for packed register k (two cells per lane), in the
// <<8 int16 domain; appends the four direction flags to the accumulators.
// Round 5: three of its differences are 32-bit subtracts (v_sub_u32: ~2.3 cycles against ~4.2 for v_pk_sub_i16,
// profiles/r05_ubench_valu_ops.txt).  They are exact on the packed halves for EVERY cell, the artefact cells of a band's
// edges included, because they never borrow: the score register only ever holds fresh scores z0 = (score + 2 (q + e)) << 8
// with q <= z0 >> 8 <= 127 (sdf_api.hip: core32_ok -- other scorings run on the general kernel), z1 = max_i(z0, a) is z0 or a
// larger non-negative value, zb = max_i(z1, b) likewise, and z3 = min_u(max_u(z1, b), cap) >= min(z1, cap) >= q << 8.  The
// other sums and differences involve u and v, which ARE negative in those cells: they keep the packed forms.
#define SDF_CORE(k)                                                     \
  {                                                                     \
    const unsigned a_ = pk_add(xt1[k], vt1[k]);                         \
    const unsigned bb_ = pk_add(Y[k], U[k]);                            \
    const unsigned z0_ = S[k];                                          \
    const unsigned z1_ = pk_maxi(z0_, a_);                              \
    const unsigned fa_ = z1_ - z0_; /* != 0 <=> a > z (signed); no borrow: z1 >= z0 >= 0 */ \
    const unsigned zb_ = pk_maxi(z1_, bb_);                             \
    const unsigned fb_ = zb_ - z1_; /* != 0 <=> b > max(z,a); no borrow */ \
    const unsigned z2_ = pk_maxu(z1_, bb_);                             \
    const unsigned z3_ = pk_minu(z2_, capv);                            \
    const unsigned un_ = pk_sub(z3_, vt1[k]);                           \
    const unsigned vn_ = pk_sub(z3_, U[k]);                             \
    const unsigned zq_ = z3_ - qv; /* no borrow: z3 >= q << 8 */         \
    const unsigned a2_ = pk_sub(a_, zq_);                               \
    const unsigned b2_ = pk_sub(bb_, zq_);                              \
    const unsigned xn_ = pk_maxi(a2_, 0u);                              \
    const unsigned yn_ = pk_maxi(b2_, 0u);                              \
    U[k] = un_;                                                         \
    V[k] = vn_;                                                         \
    X[k] = xn_;                                                         \
    Y[k] = yn_;                                                         \
    Fa[k] = shl1_or(Fa[k], pk_nonzero(fa_));                            \
    Fb[k] = shl1_or(Fb[k], pk_nonzero(fb_));                            \
    Fx[k] = shl1_or(Fx[k], pk_nonzero(xn_));                            \
    Fy[k] = shl1_or(Fy[k], pk_nonzero(yn_));                            \
  }

// fresh (score + 2(q+e)) << 8 of the two cells of a lane: two byte permutes (the window entry -> selector, selector -> scores)
#define SDF_FRESH(z, k, qraw) SDF_SCORE2(z, k, qsel_spread(qraw), has_n)

// entries of the LDS sequence windows: the whole (padded) sequence when it is short, else the window slots plus
// 1024 entries of slack (see extz2_pair.hip)
__host__ __device__ inline int wave_tcap(int tlen, int nreg) {
  const int whole = (tlen + 15) / 16 * 16 + 128 * nreg + 32, win = 128 * nreg + 1024 + 64;
  return whole < win ? whole : win;
}
__host__ __device__ inline int wave_qcap(int qlen, int nreg) {
  const int whole = qlen + 128 * nreg + 36, win = 128 * nreg + 1024 + 68;
  return whole < win ? whole : win;
}

// STREAM: the sequences do not fit the LDS windows whole (long tasks); without it the window code compiles out.
template <int NREG, bool STREAM>
__global__ __launch_bounds__(64, NREG <= 2 ? 6 : NREG <= 4 ? 4 : NREG <= 6 ? 3 : 2) void extz2_wave_kernel(const PlanTask *__restrict__ plan,
                                                        const int32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ pool, ScoreK sc,
                                                        uint8_t *__restrict__ dirbase,
                                                        sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint8_t lds[];
  constexpr int NSLOT = 128 * NREG;
  const PlanTask tk = plan[order[blockIdx.x]];
  // (a long task is a chain of dependent rows that ends the launch: its wavefront gets the SIMD before those of short
  // tasks sharing it)
  // (one level below the banded stripes' (extz2_bstripe.hip: 2 / 3): a window of at most 256 slots walks its rows in a quarter of
  // the time a stripe at the band's edge takes for the same row -- in a batch that holds both, the stripes' chain is the
  // one that ends the call)
  // (measured on the mm8-like batch of 3,000 tasks: 3 / 2 as the pair kernel 21.1 ms, 2 / 1 20.1-20.5 ms, 1 / 0 21.8-22.5 ms -- the
  // one-task kernel's own long chains then become the end of the call)
  if (tk.qlen + tk.tlen >= 16384) __builtin_amdgcn_s_setprio(2);
  else if (tk.qlen + tk.tlen >= 6144) __builtin_amdgcn_s_setprio(1);
  const int lane = threadIdx.x;
  const int qlen = tk.qlen, tlen = tk.tlen, w = tk.w;
  // Sequence windows in LDS.  Tb[i] = target position tt0 + i (16-bit codes, zero beyond the ends); W[i] = entry
  // we0 + i of the reversed query with a 32-element front pad, as byte PAIRS: entry j = QR[j-32] | QR[j-31] << 8
  // (any j is one aligned 16-bit load; N is 0x80|wild in a byte).  Short sequences fit whole; of long ones only the
  // part the band is moving through is resident and the windows are re-filled at block starts (STREAM).
  const int tcap = wave_tcap(tlen, NREG), qcap = wave_qcap(qlen, NREG);
  uint16_t *Tb = reinterpret_cast<uint16_t *>(lds);
  uint16_t *W = reinterpret_cast<uint16_t *>(lds + 2 * tcap);
  const int64_t tw_off = tk.t_word, qw_off = tk.q_word;
  int tt0_v = 0, we0_v = 0;  // window origins (always 0 without STREAM)
#define tt0 (STREAM ? tt0_v : 0)
#define we0 (STREAM ? we0_v : 0)
  auto fill_target = [&](const int from) {  // (pointers rebuilt here: re-fills are rare, registers are not)
    const uint32_t *tw = pool + tw_off, *tn = tw + (tlen + 15) / 16;
    tt0_v = from;
    for (int i = lane; i < tcap; i += 64) {
      const int t = from + i;
      Tb[i] = t < tlen ? (uint16_t)pool_code16(tw, tn, t, sc.wild) : 0;
    }
  };
  auto fill_query = [&](const int from) {
    const uint32_t *qw = pool + qw_off, *qn = qw + (qlen + 15) / 16;
    we0_v = from;
    for (int i = lane; i < qcap; i += 64) {
      const int e0 = from + i - 32, e1 = e0 + 1;  // QR indices; QR[e] = query[qlen-1-e], 0 outside
      uint32_t v0 = (e0 >= 0 && e0 < qlen) ? pool_code16(qw, qn, qlen - 1 - e0, sc.wild) : 0u;
      uint32_t v1 = (e1 >= 0 && e1 < qlen) ? pool_code16(qw, qn, qlen - 1 - e1, sc.wild) : 0u;
      W[i] = qsel_pair(v0, v1);  // (selector form: SDF_SCORE2)
    }
  };

  // ---- unpack the 2-bit / N-mask sequences into LDS ----
  bool has_n;
  {
    const uint32_t *tn = pool + tw_off + (tlen + 15) / 16, *qn = pool + qw_off + (qlen + 15) / 16;
    uint32_t n_seen = 0;
    for (int k = lane; k < (tlen + 31) / 32; k += 64) n_seen |= tn[k];
    for (int k = lane; k < (qlen + 31) / 32; k += 64) n_seen |= qn[k];
    has_n = __builtin_amdgcn_readfirstlane((int)__any(n_seen != 0)) != 0;  // wave-uniform
    fill_target(0);
    // row 0 reads entries up to qlen + NSLOT + 31: the window's top there
    fill_query(qlen + NSLOT + 36 > qcap ? qlen + NSLOT + 36 - qcap : 0);
  }
  __syncthreads();

  // ---- constants of the <<8 difference domain ----
  const unsigned qv = ((unsigned)sc.q_b << 8) * 0x00010001u;
  const unsigned capv = ((unsigned)sc.cap_b << 8) * 0x00010001u;
  const unsigned t_mis4 = (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff) * 0x01010101u, t_wild4 = (unsigned)sc.qe2_b * 0x01010101u;
  const unsigned t_delta = (unsigned)((sc.sc_match + sc.qe2_b) & 0xff) ^ (unsigned)((sc.sc_mis + sc.qe2_b) & 0xff);
  const unsigned z_wild = ((unsigned)sc.qe2_b << 8) * 0x00010001u;  // score 0, also "never written"
  unsigned one2 = 0x00010001u;  // min(x, 1) per half; opaque so that it stays one v_pk_min_u16
  SDF_OPQ(one2);

  unsigned U[NREG], V[NREG], X[NREG], Y[NREG], S[NREG], TA[NREG], TB[NREG];
  // the score tables of the lane's two slots of register k from their entries of the target window (two 16-bit codes)
  auto load_target = [&](const int k, const uint32_t two) {
    TA[k] = score_table(two & 0xffffu, t_mis4, t_delta, t_wild4);
    TB[k] = score_table(two >> 16, t_mis4, t_delta, t_wild4);
  };
  unsigned Fa[NREG], Fb[NREG], Fx[NREG], Fy[NREG];
#pragma unroll
  for (int k = 0; k < NREG; ++k) {
    U[k] = V[k] = X[k] = Y[k] = 0u;
    S[k] = z_wild;
    Fa[k] = Fb[k] = Fx[k] = Fy[k] = 0u;
    load_target(k, *reinterpret_cast<const uint32_t *>(Tb + 128 * k + 2 * lane));
  }

  const bool with_dir = !(tk.flag & SDF_FLAG_SCORE_ONLY);
  uint4 *dir = reinterpret_cast<uint4 *>(dirbase + tk.dir_off);
  const int nrow = qlen + tlen - 1;
  const int bperm_idx = ((lane + 8) & 63) * 4;

  int base = 0;
  int prev_lo = -1;
  unsigned carry_x = 0u, carry_v = 0u;  // halves shifted into slot 0 on the first row of a block
  bool zero_low = false;  // slots below the reference window still hold x,v that must read as 0
  int32_t h_top = 0, h_under = 0;  // H of the top cell / of the cell the next top cell will read
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
      const unsigned uval = r ? (((unsigned)sc.q_b << 8) << ((sr & 1) * 16)) : 0u;
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
      const unsigned vcarry = (base == 0 && r > 0) ? ((unsigned)sc.q_b << 24)
                                                   : (r == r0 ? carry_v << 16 : 0u);
      const unsigned xcarry = (base != 0 && r == r0) ? carry_x << 16 : 0u;
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
          SDF_FRESH(z, k, qc)
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
        if (r == 0) h_top = (int32_t)(uh >> 8) - 2 * sc.qe;
        else h_top = (hi0 > 0 ? h_under : h_top) + (int32_t)(uh >> 8) - sc.qe;
      }
      if (up || r == 0) {
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
      qaddr = (unsigned)(2 * tcap + 2 * (qlen - 1 - rb + base + 32 - we0 + 2 * lane));
#pragma unroll
      for (int k = 0; k < NREG; ++k) qnext[k] = *reinterpret_cast<const uint16_t *>(lds + qaddr + 256 * k);
    }
    qrow = re;
    if (STEADY && !SCALARH) hcnt += re - rb;  // every steady row takes one path step
    const unsigned vcar = base == 0 ? ((unsigned)sc.q_b << 24) : 0u;  // v carry into slot 0 (r > 0)
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
          xs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)X[0], 0x138, 0xf, 0xf, true);
          if (STEADY) vs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)V[0], 0x138, 0xf, 0xf, true);
          else vs = (unsigned)__builtin_amdgcn_update_dpp((int)vcar, (int)V[0], 0x138, 0xf, 0xf, false);
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
          SDF_FRESH(z, k, qcur[k])
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
          SDF_FRESH(z, k, qcur[k])
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
        if (STREAM && __builtin_expect(base + NSLOT > tt0 + tcap, 0)) {  // beyond the resident part of the target
          fill_target(base);
          __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < NREG; ++k)
          load_target(k, *reinterpret_cast<const uint32_t *>(Tb + (base - tt0) + 128 * k + 2 * lane));
        zero_low = false;
      }
      if (STREAM) {  // reversed-query entries this block reads (rows r0 .. r0+16, the last as a prefetch): resident?
        const int e_lo = qlen - 1 - (r0 + 16) + base + 32, e_hi = qlen - 1 - r0 + base + 32 + NSLOT - 1;
        if (__builtin_expect(e_lo < we0 || e_hi >= we0 + qcap, 0)) {
          // they move towards lower entries as the rows advance: the block's range goes to the window's top
          const int from = e_hi + 1 - qcap;
          fill_query(from < 0 ? 0 : from);
          __syncthreads();
          qrow = -1;
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
        bool special = !lean_ok || r == 0 || (r == r0 && (carry_x | carry_v) != 0u);
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
  }

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

template __global__ void extz2_wave_kernel<1, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<1, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<2, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<2, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<3, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<3, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<6, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<6, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<4, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<4, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<8, false>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);
template __global__ void extz2_wave_kernel<8, true>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK,
                                                     uint8_t *, sdf_result *);

// the windows hold the sequences whole?
bool wave_fits_whole(int qlen, int tlen, int nreg) {
  return wave_tcap(tlen, nreg) == (tlen + 15) / 16 * 16 + 128 * nreg + 32 && wave_qcap(qlen, nreg) == qlen + 128 * nreg + 36;
}

size_t wave_lds_bytes(int qlen, int tlen, int nreg) {
  return 2 * (size_t)wave_tcap(tlen, nreg) + 2 * (size_t)wave_qcap(qlen, nreg);
}

}  // namespace sdf
