// Strip kernel: full-band tasks of a few hundred to a thousand target bases, the matrix walked ROW-MAJOR by a wavefront.
//
// The window kernels sweep anti-diagonals: a full-band matrix of q x t cells takes q + t rows of a window t slots wide --
// half of the slots of a square task's window never hold a cell.  Here every lane owns a STRIP of eight target columns
// (64 lanes: a block of 512 columns) and walks it row by row, one query row per step, a step behind its left neighbour:
// at step s lane l computes the eight cells (8 l .. 8 l + 7, s - l) from the column state in its registers (u, y of the
// row above) and from x, v of its left neighbour's last cell of the same row -- which that lane computed one step
// earlier and hands over with one DPP shift.  The query bases travel the same way: lane 0 takes the base of row s, every
// other lane the one its left neighbour used a step ago.  All 64 lanes work on all but 63 of the q + 63 steps: 500 rows
// keep a wavefront 89 % busy, where the anti-diagonal sweep of a 500 x 500 task is at 50 %.
//
// Two tasks share a wavefront, task A in the low and task B in the high 16 bits of every register (packed 16-bit ALU: two
// cells per instruction).  They need not have the same geometry -- the wavefront steps through the larger matrix, and what
// a half computes beyond its own task's matrix feeds no cell inside it -- so the planner pairs neighbours of the tasks
// sorted by size; a task left over is paired with itself.  Values are the plain
// bytes of the reference's state (extern/ksw2_extz2_sse.cc:26-47,172-194): the kernel takes tame scorings only
// (match + 2 (q + e) <= 127: every byte stays in 0..127, no wrap-around, no signed / unsigned or sign-extension
// artefacts) and full bands (no cell outside the matrix feeds one inside: SURVEY 7), like the lane kernel
// (extz2_lane.hip) whose recurrence this is.  Targets of 513 .. 1024 bases take two blocks one after the other; x, v of
// column 511 wait in LDS, one word per row.  Wider tasks keep to the multi-wavefront stripe kernel (extz2_stripe.hip).
//
// Direction flags: per step and lane 8 bytes -- task A's word, task B's word: byte 0 a > z, byte 1 b > z', byte 2 x > 0,
// byte 3 y > 0, bit 7 - k of a byte = the lane's column k -- 512 contiguous bytes per step of a wavefront
// (traceback.hip: layout 6).  With four columns per lane (the chains) a byte is two steps' worth: the even step's four
// columns in its low, the odd step's in its high bits, one record per PAIR of steps -- four bits a cell like every other
// register-resident kernel (a byte a cell until the end of round 6: the flag workspace of a chain was twice what it needed).  The exact H runs along the first row (sums of u, :231 read as bytes) and down the last
// column (v): score, mte (:226-267).
#include <hip/hip_runtime.h>

#include "sdf_internal.h"

namespace sdf {

// columns per lane: 8 (a block of 512 columns per wavefront), or 4 for chains of few wavefronts -- a step of four cells
// is half as long, and a chain of blocks runs at the pace of its steps
constexpr int kStripMaxT = 512;        // widest target of the one-wavefront kernel: one block of 8 columns per lane
constexpr int kStripChainMaxT = 65536;  // ... of a chain of wavefronts, one per block (extz2_strip_chain_kernel): the block index
                                       // of a launch entry has eight bits (256 blocks of 256 columns); the stage's tasks end at 60 kb

__host__ __device__ inline int strip_blocks(int tlen, int cols = 8) { return (tlen + 64 * cols - 1) / (64 * cols); }
__host__ __device__ inline int strip_records(int qlen, int cols) {  // records per block: one per step, or per pair of steps
  return cols == 4 ? (qlen + 64) >> 1 : qlen + 63;
}
__host__ __device__ inline size_t strip_dir_bytes(int qlen, int tlen, int cols = 8, bool solo = false) {
  return (size_t)strip_blocks(tlen, cols) * (size_t)strip_records(qlen, cols) * (solo ? 256 : 512);
}
__host__ __device__ inline size_t strip_lds_bytes(int qlen, int tlen) {
  return strip_blocks(tlen) > 1 ? ((size_t)(qlen + 66) * 4 + 15) & ~(size_t)15 : 16;
}

// packed code of position k of tasks A | B << 16: 0..3, N = 4; 0 beyond a task's end
__device__ __forceinline__ unsigned strip_code2(const uint32_t *wa, const uint32_t *na, int lena, const uint32_t *wb,
                                                const uint32_t *nb, int lenb, int k) {
  unsigned ca = 0u, cb = 0u;
  if (k < lena) ca = ((na[k >> 5] >> (k & 31)) & 1u) ? 4u : ((wa[k >> 4] >> ((k & 15) * 2)) & 3u);
  if (k < lenb) cb = ((nb[k >> 5] >> (k & 31)) & 1u) ? 4u : ((wb[k >> 4] >> ((k & 15) * 2)) & 3u);
  return ca | (cb << 16);
}

template <int C>
__device__ __forceinline__ unsigned strip_pick(const unsigned (&v)[C], const int k) {  // k: wave-uniform, 0 .. C - 1
  const unsigned a = (k & 1) ? v[1] : v[0], b = (k & 1) ? v[3] : v[2];
  const unsigned e = (k & 2) ? b : a;
  if (C == 4) return e;
  const unsigned c = (k & 1) ? v[C - 3] : v[C - 4], d = (k & 1) ? v[C - 1] : v[C - 2];
  const unsigned f = (k & 2) ? d : c;
  return (k & 4) ? f : e;
}

__device__ __forceinline__ int strip_wave_sum(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// what a task's half of the wavefront keeps of the exact H (32-bit, outside the packed registers)
struct StripH {
  int sum0 = 0;                       // sum of u over the lane's columns of row 0
  int hrel = 0, best = 0, bestj = 0;  // H of the last column relative to its row 0; the first maximum and its row
};

// One block of 512 columns, all rows.  Tasks A and B need not be equal: the wavefront steps through max(qlen) rows of
// max(tlen) columns, and what a task's half computes beyond its own matrix feeds nothing inside it (cells depend on
// cells above and to the left only); each half follows its own last column and its own last row.
template <bool HASN, bool CHAIN, int C>
__device__ __forceinline__ bool strip_steps(const int blk, const int nblk, const int qa, const int ta, const int qb, const int tb,
                                            const int lane, const uint32_t *twa, const uint32_t *tna, const uint32_t *twb,
                                            const uint32_t *tnb, const uint32_t *qwa, const uint32_t *qna, const uint32_t *qwb,
                                            const uint32_t *qnb, const int gq, const int qe, const unsigned ZM2, const unsigned ZD2,
                                            const unsigned ZW2, const unsigned CAP2, uint32_t *edge, uint2 *dir,
                                            const bool with_dir_a, const bool with_dir_b, StripH &ha, StripH &hb,
                                            const uint32_t *edge_in = nullptr, uint32_t *edge_out = nullptr,
                                            const sdf_result *mark = nullptr, const int spin_cap = 0, int *sums = nullptr,
                                            const bool solo = false) {
  const unsigned Q2 = (unsigned)gq * 0x00010001u;
  unsigned one2 = 0x00010001u, two2 = 0x00020002u;
  SDF_OPQ(one2);
  SDF_OPQ(two2);
  const int qmax = qa > qb ? qa : qb;
  constexpr int BW = 64 * C;  // columns of a block
  const int c0 = blk * BW + lane * C;
  // A cell's score + 2 (q + e) is ONE byte permute: per column a table of four bytes per task -- the score against query
  // base 0..3 -- and the row's selector: byte 0 = task A's query base (picks a byte of TA), byte 2 = 4 + task B's (a byte of
  // TB), bytes 1 and 3 = 0x0c (zero).  (Until the end of round 6: xor, min, multiply-add per cell -- a launch of chained
  // strips issues a vector instruction on 92 % of its SIMD cycles, profiles/r06_chain_pmc.txt: what counts is their number.)
  // An N on either side (HASN) is patched in afterwards: the table's answer for it is arbitrary.
  unsigned U[C], Y[C], TA[C], TB[C], NT[HASN ? C : 1];
  const unsigned zm1 = ZM2 & 0xffu, zx1 = (ZM2 + ZD2) & 0xffu;  // match, mismatch (+ 2 (q + e))
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const unsigned tc = strip_code2(twa, tna, ta, twb, tnb, tb, c0 + k);
    const unsigned ca = tc & 0xffffu, cb = tc >> 16;
    TA[k] = (zx1 * 0x01010101u) ^ (ca < 4u ? (zm1 ^ zx1) << (8u * ca) : 0u);
    TB[k] = (zx1 * 0x01010101u) ^ (cb < 4u ? (zm1 ^ zx1) << (8u * cb) : 0u);
    if (HASN) NT[k] = (tc >> 2) & 0x00010001u;
    U[k] = (c0 + k) ? Q2 : 0u;  // (:121: u = q beyond the first column, y = 0 above the first row)
    Y[k] = 0u;
  }
  // the lane's columns inside each target; each target's last column, if it lies in this block (wave-uniform)
  const int nva = ta - c0 < 0 ? 0 : ta - c0 > C ? C : ta - c0, nvb = tb - c0 < 0 ? 0 : tb - c0 > C ? C : tb - c0;
  const int cla = ta - 1 - blk * BW, clb = tb - 1 - blk * BW;
  const bool last_a = cla >= 0 && cla < BW, last_b = clb >= 0 && clb < BW;
  const int kca = last_a ? (cla % C) : 0, kcb = last_b ? (clb % C) : 0, lla = cla / C, llb = clb / C;
  const bool more = blk + 1 < nblk;
  const int nstep = qmax + 63;
  constexpr bool PK = C == 4;  // two steps' flags per record
  const int nrec = strip_records(qmax, C);
  const int nloop = PK ? 2 * nrec : nstep;  // (PK: an even number of steps -- the last one may only write the record out)
  unsigned hold_a = 0u, hold_b = 0u;        // PK: the even step's words
  unsigned xo = 0u, vo = 0u, qc = 0x0c040c00u, vcap = 0u;
  unsigned qrot = 0u;  // the query bases of sixteen rows, rotating (below)
  unsigned e_next = (!CHAIN && blk) ? edge[0] : 0u;
  unsigned e16 = 0u;  // CHAIN: the edge words of rows (s & ~15) + lane, lanes 0..15
  uint2 *drow = dir + (size_t)blk * nrec * 64 + lane;
  for (int s = 0; s < nloop; ++s) {
    if (CHAIN && s == 64) {
      // every lane has passed row 0: this block's share of the first row's sum of u, published before any edge word of
      // a row beyond 0 leaves lane 63 (the blocks holding the tasks' last columns read the totals at their very end)
      const int s0a = strip_wave_sum(ha.sum0), s0b = strip_wave_sum(hb.sum0);
      if (lane == 0) {
        if (s0a) atomicAdd(sums, s0a);
        if (s0b) atomicAdd(sums + 1, s0b);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    if (CHAIN && blk && (s & 15) == 0) {
      // the sixteen edge words the next sixteen steps read: written (tag bit 7) by the wavefront of the block to the left,
      // which runs 64 steps ahead once it is under way -- and may not have started when this one does
      const int row = s + (lane & 15);
      int spins = 0;
      for (;;) {
        e16 = row < qmax ? ld_agent(edge_in + row) : 0x80u;
        if (!__builtin_amdgcn_readfirstlane((int)__any((e16 & 0x80u) == 0u))) break;
        if (++spins >= spin_cap || ((spins & 63) == 63 && stripe_abandoned(mark))) return false;
        if (s) __builtin_amdgcn_s_sleep(4);
        else __builtin_amdgcn_s_sleep(64);
      }
    }
    // lane 0 takes the query bases of row s; the others those their left neighbour had (lanes run a step apart).
    // Round 5: the bases of the next sixteen rows are unpacked ONCE per sixteen steps, lane i holding row (s & ~15) + i's
    // (A | B << 16), and the register rotates one lane per step, so that lane 0 always holds the current row's: one DPP
    // a step where the shifts, masks and selects of the two tasks' words were twelve of a step's ~190 instructions.
    if ((s & 15) == 0) {
      const int jr = s + (lane & 15);
      unsigned ca = 0u, cb = 0u;
      if (jr < qa) ca = ((qna[jr >> 5] >> (jr & 31)) & 1u) ? 4u : ((qwa[jr >> 4] >> ((jr & 15) * 2)) & 3u);
      if (jr < qb) cb = ((qnb[jr >> 5] >> (jr & 31)) & 1u) ? 4u : ((qwb[jr >> 4] >> ((jr & 15) * 2)) & 3u);
      qrot = ca | ((cb + 4u) << 16) | 0x0c000c00u;  // (the selector of the score tables, above)
    } else {
      qrot = (unsigned)__builtin_amdgcn_mov_dpp((int)qrot, 0x134, 0xf, 0xf, false);  // wave_rol:1
    }
    qc = (unsigned)__builtin_amdgcn_update_dpp((int)qrot, (int)qc, 0x138, 0xf, 0xf, false);
    // x, v left of the strip: the left neighbour's last cell of this row; lane 0: the matrix border (:120: x = 0, v = q
    // below the first row) or, from the second block on, the last column of the block before
    unsigned e_cur = e_next;
    if (CHAIN) e_cur = (unsigned)__builtin_amdgcn_readlane((int)e16, s & 15);
    else if (blk) e_next = edge[s + 1];
    const unsigned xb = blk ? ((e_cur & 0x7fu) | (e_cur & 0xff0000u)) : 0u;
    const unsigned vb = blk ? (((e_cur >> 8) & 0xffu) | ((e_cur >> 8) & 0xff0000u)) : (s ? Q2 : 0u);
    unsigned x = (unsigned)__builtin_amdgcn_update_dpp((int)xb, (int)xo, 0x138, 0xf, 0xf, false);
    unsigned v = (unsigned)__builtin_amdgcn_update_dpp((int)vb, (int)vo, 0x138, 0xf, 0xf, false);
    const int j = s - lane;  // this lane's row
    unsigned wa = 0u, wb = 0u;  // the step's flag words (0 outside the matrix)
    if (j >= 0 && j < qmax) {
      unsigned Fa = 0u, Fb = 0u, Fx = 0u, Fy = 0u;
      unsigned VN[C];
      // (HASN) the row's base is an N: selector byte 0 = 4, byte 2 = 8
      const unsigned nq = HASN ? ((qc >> 2) & 1u) | ((qc >> 3) & 0x00010000u) : 0u;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        // score + 2 (q + e): match / mismatch by the bases, 2 (q + e) with an N on either side (:124-138)
        unsigned z = __builtin_amdgcn_perm(TB[k], TA[k], qc);
        if (HASN) {
          const unsigned nn = NT[k] | nq;
          z = pk_mad(nn, pk_sub(ZW2, z), z);
        }
        // Round 5: the sums and differences are plain 32-bit adds and subtracts.  gfx950 issues v_add_u32 / v_sub_u32 in ~2.3
        // cycles and every v_pk_* in ~4.2 (profiles/r05_ubench_valu_ops.txt), and on THIS kernel's values they are exact on both
        // halves: every byte of the state is in 0..127 (tame scoring, full band: the header), so no sum carries out of a half,
        // and every difference taken here is of a value and something it was maximised over or that the recurrence keeps below
        // it (z1 - z, z2 - z1; u = z - v >= 0, v = z - u >= 0; z - q >= 0: mismatch + q + 2 e >= 0 is part of strip_ok), so none
        // borrows.  The two that can go negative, a - (z - q) and b - (z - q), are saturating packed subtracts (max(. - ., 0) in one).
        const unsigned uo = U[k];
        const unsigned a = x + v, b = Y[k] + uo;
        const unsigned z1 = pk_maxu(z, a);  // ties: diagonal before E before F (:173-178)
        const unsigned z2 = pk_maxu(z1, b);
        const unsigned fa = pk_minu(z1 - z, one2), fb = pk_minu(z2 - z1, one2);
        const unsigned z3 = pk_minu(z2, CAP2);
        const unsigned un = z3 - v, vn = z3 - uo;
        const unsigned zq = z3 - Q2;
        x = pk_subsat_u(a, zq);  // max(a - (z - q), 0): one saturating subtract
        const unsigned yn = pk_subsat_u(b, zq);
        Fa = pk_mad(Fa, two2, fa);
        Fb = pk_mad(Fb, two2, fb);
        Fx = pk_mad(Fx, two2, pk_minu(x, one2));
        Fy = pk_mad(Fy, two2, pk_minu(yn, one2));
        v = vn;
        VN[k] = vn;
        U[k] = un;
        Y[k] = yn;
      }
      // v of a target's last column, in the lane that holds it (the column index within a lane is wave-uniform)
      if (last_a) vcap = (vcap & 0xffff0000u) | (strip_pick(VN, kca) & 0xffffu);
      if (last_b) vcap = (vcap & 0xffffu) | (strip_pick(VN, kcb) & 0xffff0000u);
      xo = x;
      vo = v;
      if (more && lane == 63) {
        const unsigned ew = __builtin_amdgcn_perm(vo, xo, 0x06020400u);  // bytes x_A, v_A, x_B, v_B (one permute)
        if (CHAIN) st_agent(edge_out + j, ew | 0x80u);  // (x <= 127: bit 7 tags the word as written)
        else edge[j] = ew;
      }
      {
        // byte 0 of the four accumulators' low halves -> task A's word, of their high halves -> task B's (an accumulator holds
        // at most C <= 8 bits per half): four byte permutes instead of sixteen shifts, masks and ors
        const unsigned ab = __builtin_amdgcn_perm(Fb, Fa, 0x06020400u), xy = __builtin_amdgcn_perm(Fy, Fx, 0x06020400u);
        wa = __builtin_amdgcn_perm(xy, ab, 0x05040100u), wb = __builtin_amdgcn_perm(xy, ab, 0x07060302u);
        if (!PK) {
          if (solo) {  // (a task without a partner: records of one word)
            if (with_dir_a) reinterpret_cast<uint32_t *>(dir)[((size_t)blk * nrec + s) * 64 + lane] = wa;
          } else if (with_dir_a && with_dir_b) {
            drow[(size_t)s * 64] = make_uint2(wa, wb);
          } else if (with_dir_a) {
            drow[(size_t)s * 64].x = wa;
          } else if (with_dir_b) {
            drow[(size_t)s * 64].y = wb;
          }
        }
      }
      // (the three blocks below under UNIFORM conditions of their own: as lane predicates alone the compiler runs their
      // instructions through every step with EXEC empty -- fifteen of a step's ~170 in a block that holds no last column)
      if (s < 64) {  // (row 0 passes lane l at step l)
        if (j == 0) {  // exact H along the first row: the sum of u (:231, read as bytes) over the lane's columns
#pragma unroll
          for (int k = 0; k < C; ++k) {
            if (k < nva) ha.sum0 += (int)(U[k] & 0xffffu);
            if (k < nvb) hb.sum0 += (int)(U[k] >> 16);
          }
        }
      }
      // ... and down the last column (v): ascending rows, the first maximum stays (:252)
      if (last_a) {
        if (lane == lla && j < qa) {
          ha.hrel = j ? ha.hrel + (int)(vcap & 0xffffu) - qe : 0;
          if (j == 0 || ha.hrel > ha.best) {
            ha.best = ha.hrel;
            ha.bestj = j;
          }
        }
      }
      if (last_b) {
        if (lane == llb && j < qb) {
          hb.hrel = j ? hb.hrel + (int)(vcap >> 16) - qe : 0;
          if (j == 0 || hb.hrel > hb.best) {
            hb.best = hb.hrel;
            hb.bestj = j;
          }
        }
      }
    }
    if (PK) {
      if (s & 1) {
        if (j >= 0 && j <= qmax) {  // (this step's row or the even step's, j - 1, lies inside the matrix)
          const unsigned pa = hold_a | (wa << 4), pb = hold_b | (wb << 4);
          const size_t r = (size_t)(s >> 1) * 64;
          if (solo) {
            if (with_dir_a) reinterpret_cast<uint32_t *>(dir)[((size_t)blk * nrec + (s >> 1)) * 64 + lane] = pa;
          } else if (with_dir_a && with_dir_b) {
            drow[r] = make_uint2(pa, pb);
          } else if (with_dir_a) {
            drow[r].x = pa;
          } else if (with_dir_b) {
            drow[r].y = pb;
          }
        }
      } else {
        hold_a = wa;
        hold_b = wb;
      }
    }
  }
  return true;
}

__global__ __launch_bounds__(64) void extz2_strip_kernel(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ pool, ScoreK sc,
                                                         uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res) {
  extern __shared__ __align__(16) uint32_t strip_edge[];
  const int ia = order[2 * blockIdx.x], ib = order[2 * blockIdx.x + 1];
  const PlanTask tka = plan[ia], tkb = plan[ib];  // task A in the low halves, B in the high ones (A again without a partner)
  const int lane = threadIdx.x;
  const int qa = tka.qlen, ta = tka.tlen, qb = tkb.qlen, tb = tkb.tlen;
  const uint32_t *twa = pool + tka.t_word, *tna = twa + (ta + 15) / 16;
  const uint32_t *qwa = pool + tka.q_word, *qna = qwa + (qa + 15) / 16;
  const uint32_t *twb = pool + tkb.t_word, *tnb = twb + (tb + 15) / 16;
  const uint32_t *qwb = pool + tkb.q_word, *qnb = qwb + (qb + 15) / 16;
  bool has_n;
  {
    uint32_t seen = 0;
    for (int k = lane; k < (ta + 31) / 32; k += 64) seen |= tna[k];
    for (int k = lane; k < (tb + 31) / 32; k += 64) seen |= tnb[k];
    for (int k = lane; k < (qa + 31) / 32; k += 64) seen |= qna[k];
    for (int k = lane; k < (qb + 31) / 32; k += 64) seen |= qnb[k];
    has_n = __builtin_amdgcn_readfirstlane((int)__any(seen != 0)) != 0;
  }
  const int gq = sc.q, qe = sc.qe;
  const int zm = (int)(int8_t)sc.sc_match + 2 * qe, zmis = (int)(int8_t)sc.sc_mis + 2 * qe;
  const unsigned ZM2 = (unsigned)zm * 0x00010001u, ZD2 = ((unsigned)(zmis - zm) & 0xffffu) * 0x00010001u;
  const unsigned ZW2 = (unsigned)(2 * qe) * 0x00010001u, CAP2 = ZM2;
  const bool with_dir_a = !(tka.flag & SDF_FLAG_SCORE_ONLY), with_dir_b = ib != ia && !(tkb.flag & SDF_FLAG_SCORE_ONLY);
  uint2 *dir = reinterpret_cast<uint2 *>(dirbase + tka.dir_off);  // (task B's words are the odd ones of the same records)
  const int nblk = strip_blocks(ta > tb ? ta : tb);
  StripH ha, hb;
  for (int blk = 0; blk < nblk; ++blk) {
    if (has_n)
      strip_steps<true, false, 8>(blk, nblk, qa, ta, qb, tb, lane, twa, tna, twb, tnb, qwa, qna, qwb, qnb, gq, qe, ZM2, ZD2, ZW2, CAP2,
                               strip_edge, dir, with_dir_a, with_dir_b, ha, hb, nullptr, nullptr, nullptr, 0, nullptr, ib == ia);
    else
      strip_steps<false, false, 8>(blk, nblk, qa, ta, qb, tb, lane, twa, tna, twb, tnb, qwa, qna, qwb, qnb, gq, qe, ZM2, ZD2, ZW2, CAP2,
                                strip_edge, dir, with_dir_a, with_dir_b, ha, hb, nullptr, nullptr, nullptr, 0, nullptr, ib == ia);
  }
  // H of the last cell of row 0: u(0,0) - 2 (q + e) (:249), every further cell of the row + its u - (q + e) (:231)
  const int row0a = strip_wave_sum(ha.sum0) - (ta - 1) * qe - 2 * qe, row0b = strip_wave_sum(hb.sum0) - (tb - 1) * qe - 2 * qe;
  const int lla = ((ta - 1) & 511) >> 3, llb = ((tb - 1) & 511) >> 3;
  const int fa = __builtin_amdgcn_readlane(ha.hrel, lla), fb = __builtin_amdgcn_readlane(hb.hrel, llb);
  const int ba = __builtin_amdgcn_readlane(ha.best, lla), bb = __builtin_amdgcn_readlane(hb.best, llb);
  const int ja = __builtin_amdgcn_readlane(ha.bestj, lla), jb = __builtin_amdgcn_readlane(hb.bestj, llb);
  if (lane < 2 && (lane == 0 || ib != ia)) {
    const int tl = lane ? tb : ta;
    sdf_result o;
    o.score = lane ? row0b + fb : row0a + fa;  // (the last row's value, :255)
    o.max = 0;
    o.max_q = o.max_t = -1;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = lane ? row0b + bb : row0a + ba;
    o.mte_q = (lane ? jb : ja) + (tl - 1) - ((tl - 1) | 15);  // (r - en with the reference's block-rounded en, :262)
    o.zdropped = 0;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    res[lane ? tkb.out_idx : tka.out_idx] = o;
  }
}

// ---- wide targets: a chain of wavefronts, one per block of 512 columns ----------------------------------------------
// Targets of 513 .. 65536 bases: block b of a pair of tasks is the work of one wavefront (its own workgroup), 64 steps
// behind the wavefront of block b - 1, which hands x, v of its last column over through HBM: one word per row for both
// tasks, bit 7 set when written (x <= 127: the bit is free), relaxed atomics of agent scope, fetched sixteen rows at a
// time.  Forward progress as in extz2_stripe.hip: a pair's blocks share an XCD (workgroup index mod 8) and are listed
// in ascending order, so the wavefront a block waits for has a smaller workgroup index on the same XCD -- resident or
// finished; a wait that runs out of polls abandons both tasks (stripe_abandon) and the batch call runs them again.
// A block's partial exact-H values (the first row's sum of u; the last column's running H) are added up through three
// words per task behind the edge columns.
__host__ __device__ inline size_t strip_chain_sync_bytes(int qmax, int tmax, int cols) {
  return ((size_t)(strip_blocks(tmax, cols) - 1) * (size_t)(qmax + 64) * 4 + 64 + 255) & ~(size_t)255;
}

__global__ __launch_bounds__(64) void strip_chain_init_kernel(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                                              uint8_t *__restrict__ dirbase) {
  const int32_t entry = order[blockIdx.x];
  const int blk = (int)((uint32_t)entry >> 24);
  const PlanTask tka = plan[entry & 0xffffff], tkb = plan[tka.zdrop];
  const int qmax = tka.qlen > tkb.qlen ? tka.qlen : tkb.qlen, tmax = tka.tlen > tkb.tlen ? tka.tlen : tkb.tlen;
  const int nblk = strip_blocks(tmax, tka.nreg);  // (nreg: columns per lane)
  if (blk >= nblk) return;
  uint32_t *sync = reinterpret_cast<uint32_t *>(dirbase + tka.dir_off +
                                                (int64_t)strip_dir_bytes(qmax, tmax, tka.nreg, tka.zdrop == (entry & 0xffffff)));
  if (blk == 0)
    for (int k = threadIdx.x; k < 16; k += 64) sync[k] = 0u;  // the row-0 sums of both tasks
  if (blk + 1 < nblk) {
    uint32_t *col = sync + 16 + (size_t)blk * (qmax + 64);
    for (int k = threadIdx.x; k < qmax + 64; k += 64) col[k] = 0u;
  }
}

template <int C>
__global__ __launch_bounds__(64) void extz2_strip_chain_kernel(const PlanTask *__restrict__ plan, const int32_t *__restrict__ order,
                                                               const uint32_t *__restrict__ pool, ScoreK sc,
                                                               uint8_t *__restrict__ dirbase, sdf_result *__restrict__ res,
                                                               unsigned long long *__restrict__ gave_up, const int spin_cap,
                                                               unsigned *__restrict__ claim) {
  const int32_t entry = stripe_claim(order, claim);
  const int blk = (int)((uint32_t)entry >> 24);
  const int ia = entry & 0xffffff;
  const PlanTask tka = plan[ia];
  const int ib = tka.zdrop;  // (the partner's plan entry; the task itself without one)
  const PlanTask tkb = plan[ib];
  const int lane = threadIdx.x;
  const int qa = tka.qlen, ta = tka.tlen, qb = tkb.qlen, tb = tkb.tlen;
  const int qmax = qa > qb ? qa : qb, tmax = ta > tb ? ta : tb;
  const int nblk = strip_blocks(tmax, C);
  if (blk >= nblk) return;  // (a padding entry of the launch order)
  place_note();
  if (qmax + tmax >= 8192) __builtin_amdgcn_s_setprio(2);  // (long chains first, as in the stripe kernels)
  const uint32_t *twa = pool + tka.t_word, *tna = twa + (ta + 15) / 16;
  const uint32_t *qwa = pool + tka.q_word, *qna = qwa + (qa + 15) / 16;
  const uint32_t *twb = pool + tkb.t_word, *tnb = twb + (tb + 15) / 16;
  const uint32_t *qwb = pool + tkb.q_word, *qnb = qwb + (qb + 15) / 16;
  bool has_n;
  {
    uint32_t seen = 0;
    for (int k = lane; k < (ta + 31) / 32; k += 64) seen |= tna[k];
    for (int k = lane; k < (tb + 31) / 32; k += 64) seen |= tnb[k];
    for (int k = lane; k < (qa + 31) / 32; k += 64) seen |= qna[k];
    for (int k = lane; k < (qb + 31) / 32; k += 64) seen |= qnb[k];
    has_n = __builtin_amdgcn_readfirstlane((int)__any(seen != 0)) != 0;
  }
  const int gq = sc.q, qe = sc.qe;
  const int zm = (int)(int8_t)sc.sc_match + 2 * qe, zmis = (int)(int8_t)sc.sc_mis + 2 * qe;
  const unsigned ZM2 = (unsigned)zm * 0x00010001u, ZD2 = ((unsigned)(zmis - zm) & 0xffffu) * 0x00010001u;
  const unsigned ZW2 = (unsigned)(2 * qe) * 0x00010001u, CAP2 = ZM2;
  const bool with_dir_a = !(tka.flag & SDF_FLAG_SCORE_ONLY), with_dir_b = ib != ia && !(tkb.flag & SDF_FLAG_SCORE_ONLY);
  uint2 *dir = reinterpret_cast<uint2 *>(dirbase + tka.dir_off);
  uint32_t *sync = reinterpret_cast<uint32_t *>(dirbase + tka.dir_off + (int64_t)strip_dir_bytes(qmax, tmax, C, ib == ia));
  int *sums = reinterpret_cast<int *>(sync);  // [0] row-0 sum of u of task A, [1] of task B (all blocks add theirs)
  uint32_t *cols = sync + 16;
  const uint32_t *edge_in = cols + (size_t)(blk ? blk - 1 : 0) * (qmax + 64);
  uint32_t *edge_out = cols + (size_t)blk * (qmax + 64);
  StripH ha, hb;
  const sdf_result *mark = res + tka.out_idx;
  const bool ok = has_n ? strip_steps<true, true, C>(blk, nblk, qa, ta, qb, tb, lane, twa, tna, twb, tnb, qwa, qna, qwb, qnb, gq, qe, ZM2,
                                                  ZD2, ZW2, CAP2, nullptr, dir, with_dir_a, with_dir_b, ha, hb, edge_in, edge_out,
                                                  mark, spin_cap, sums, ib == ia)
                        : strip_steps<false, true, C>(blk, nblk, qa, ta, qb, tb, lane, twa, tna, twb, tnb, qwa, qna, qwb, qnb, gq, qe, ZM2,
                                                   ZD2, ZW2, CAP2, nullptr, dir, with_dir_a, with_dir_b, ha, hb, edge_in, edge_out,
                                                   mark, spin_cap, sums, ib == ia);
  if (!ok) {  // (the batch call runs both tasks again on another kernel)
    stripe_abandon(gave_up, res + tka.out_idx, tka.out_idx, lane);
    if (ib != ia) stripe_abandon(gave_up, res + tkb.out_idx, tkb.out_idx, lane);
    return;
  }
  // (every block has added its share of the first row's sums of u at its step 64; the blocks holding a task's last
  // column read the totals now, after all their rows -- rows whose edge words left the blocks to their left behind
  // those blocks' own additions)
  const int bla = (ta - 1) / (64 * C), blb = (tb - 1) / (64 * C);  // the blocks with the tasks' last columns
  if (blk != bla && blk != blb) return;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  const int lla = ((ta - 1) % (64 * C)) / C, llb = ((tb - 1) % (64 * C)) / C;
  const int fa = __builtin_amdgcn_readlane(ha.hrel, lla), fb = __builtin_amdgcn_readlane(hb.hrel, llb);
  const int ba = __builtin_amdgcn_readlane(ha.best, lla), bb = __builtin_amdgcn_readlane(hb.best, llb);
  const int ja = __builtin_amdgcn_readlane(ha.bestj, lla), jb = __builtin_amdgcn_readlane(hb.bestj, llb);
  if (lane < 2 && (lane == 0 ? blk == bla : (ib != ia && blk == blb))) {
    const int tl = lane ? tb : ta;
    const int row0 = ld_agent(sums + lane) - (tl - 1) * qe - 2 * qe;
    sdf_result o;
    o.score = row0 + (lane ? fb : fa);
    o.max = 0;
    o.max_q = o.max_t = -1;
    o.mqe = SDF_NEG_INF;
    o.mqe_t = -1;
    o.mte = row0 + (lane ? bb : ba);
    o.mte_q = (lane ? jb : ja) + (tl - 1) - ((tl - 1) | 15);
    o.zdropped = 0;
    o.n_cigar = 0;
    o.cigar_off = 0;
    o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
    // (n_cigar is where a block that gave up marks the task, -1: the blocks to the right of this one watch task A's record
    // only and may have marked both tasks already -- the record goes out around that field, which the batch's reset left 0,
    // so that the mark stays and nobody appends the task to the give-up list a second time)
    sdf_result *dst = res + (lane ? tkb.out_idx : tka.out_idx);
    const int keep = __hip_atomic_load(&dst->n_cigar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.n_cigar = keep == -1 ? -1 : 0;
    const int32_t *src = reinterpret_cast<const int32_t *>(&o);
    int32_t *out = reinterpret_cast<int32_t *>(dst);
    constexpr int at = offsetof(sdf_result, n_cigar) / 4;
#pragma unroll
    for (int f = 0; f < (int)(sizeof(sdf_result) / 4); ++f)
      if (f != at) out[f] = src[f];
  }
}

template __global__ void extz2_strip_chain_kernel<8>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                     sdf_result *, unsigned long long *, int, unsigned *);
template __global__ void extz2_strip_chain_kernel<4>(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                                     sdf_result *, unsigned long long *, int, unsigned *);

}  // namespace sdf
