/*
 * TEST INFRASTRUCTURE ONLY -- see extz2_oracle.h.  Never linked into the product.
 *
 * Scalar restatement of the reference's anti-diagonal, difference-form affine-gap DP
 * (Suzuki-Kasahara formulation as vendored by the reference in
 * extern/ksw2_extz2_sse.cc:23-298) including the parts of its behaviour that are artefacts
 * of the 16-byte SIMD implementation but are observable in its output:
 *
 *   - all state lives in one zero-initialised byte arena laid out  u|v|x|y|s|sf|qr
 *     (ksw2_extz2_sse.cc:83-85); reads/writes past a region land in the next one;
 *   - the band of each anti-diagonal is widened to whole 16-cell blocks (:115) and the
 *     widened cells are really computed, from whatever the arena holds;
 *   - match/mismatch scores are refreshed in 16-cell strides starting at the logical band
 *     start (:124-138), so they can run past the band end and are stale below its start;
 *   - the carry-in bytes of the first block are sign-extended 32-bit values OR-ed into the
 *     low four lanes (:145-146 with :29,:33);
 *   - the exact-H arg-max is found in four interleaved lanes merged in lane order (:226-258).
 *
 * Arithmetic is done on uint8_t with explicit int8_t reinterpretation so wrap-around is
 * well defined.
 */
#include "extz2_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int8_t s8(uint8_t b) { return (int8_t)b; }
static inline uint8_t smax8(uint8_t a, uint8_t b) { return s8(a) > s8(b) ? a : b; }

/* CIGAR builder: run-length merge of equal adjacent ops (extern/ksw2.h:98-111). */
typedef struct {
  uint32_t *v;
  int64_t n, cap;
} cigbuf;

static void cig_push(cigbuf *c, uint32_t op, int64_t len) {
  if (c->n > 0 && (c->v[c->n - 1] & 0xfu) == op) {
    c->v[c->n - 1] += (uint32_t)(len << 4);
    return;
  }
  if (c->n == c->cap) {
    c->cap = c->cap ? c->cap * 2 : 4;
    c->v = (uint32_t *)realloc(c->v, (size_t)c->cap * sizeof(uint32_t));
    if (!c->v) abort();
  }
  c->v[c->n++] = (uint32_t)(len << 4) | op;
}

/* Traceback over the per-cell direction bytes (extern/ksw2.h:117-151, rotated layout). */
static void traceback(const uint8_t *dir, const int32_t *row_lo, const int32_t *row_hi,
                      int64_t row_stride, int i0, int j0, int rev, cigbuf *cg) {
  int64_t i = i0, j = j0;
  int state = 0;
  while (i >= 0 && j >= 0) {
    int64_t r = i + j;
    int forced = -1;
    uint32_t d = 0;
    if (i < row_lo[r]) forced = 2;
    if (i > row_hi[r]) forced = 1;
    if (forced < 0) d = dir[r * row_stride + (i - row_lo[r])];
    if (state == 0) state = (int)(d & 7u);
    else if (!((d >> (state + 2)) & 1u)) state = 0;
    if (state == 0) state = (int)(d & 7u);
    if (forced >= 0) state = forced;
    if (state == 0) { cig_push(cg, 0, 1); --i; --j; }
    else if (state == 1 || state == 3) { cig_push(cg, 2, 1); --i; }
    else { cig_push(cg, 1, 1); --j; }
  }
  if (i >= 0) cig_push(cg, 2, i + 1);
  if (j >= 0) cig_push(cg, 1, j + 1);
  if (!rev) {
    for (int64_t a = 0, b = cg->n - 1; a < b; ++a, --b) {
      uint32_t t = cg->v[a];
      cg->v[a] = cg->v[b];
      cg->v[b] = t;
    }
  }
}

/* running maximum / z-drop test (extern/ksw2.h:161-177, rotated coordinates) */
static int track_max(sdfo_result *ez, int32_t H, int r, int t, int zdrop, int e) {
  if (H > (int32_t)ez->max) {
    ez->max = (uint32_t)H & 0x7fffffffu;
    ez->max_t = t;
    ez->max_q = r - t;
  } else if (t >= ez->max_t && r - t >= ez->max_q) {
    int tl = t - ez->max_t, ql = (r - t) - ez->max_q;
    int l = tl > ql ? tl - ql : ql - tl;
    if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {
      ez->zdropped = 1;
      return 1;
    }
  }
  return 0;
}

void sdfo_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                const int8_t *mat, int gapo, int gape, int w, int zdrop, int flag,
                sdfo_result *ez) {
  const int8_t q = (int8_t)gapo, e = (int8_t)gape; /* narrowed at the ABI (ksw2.h:50) */
  const int qe = q + e;
  const int want_cigar = !(flag & SDFO_SCORE_ONLY);
  const int approx = !!(flag & SDFO_APPROX_MAX);

  ez->max = 0; ez->zdropped = 0;
  ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
  ez->score = ez->mqe = ez->mte = SDFO_NEG_INF;
  ez->n_cigar = 0; ez->cigar = 0;
  if (m <= 0 || qlen <= 0 || tlen <= 0) return;

  const uint8_t q_b = (uint8_t)q;
  const uint8_t qe2_b = (uint8_t)((q + e) * 2);
  const uint8_t cap_b = (uint8_t)(mat[0] + (q + e) * 2);
  const uint8_t wild = (uint8_t)(m - 1);
  const uint8_t sc_match = (uint8_t)mat[0], sc_mis = (uint8_t)mat[1];

  if (w < 0) w = tlen > qlen ? tlen : qlen;
  const int tblk = (tlen + 15) / 16, qblk = (qlen + 15) / 16;
  int ncol = qlen < tlen ? qlen : tlen;
  ncol = ((ncol < w + 1 ? ncol : w + 1) + 15) / 16 + 1;
  int min_sc = mat[1];
  for (int k = 1; k < m * m; ++k) if (mat[k] < min_sc) min_sc = mat[k];
  if (-min_sc > 2 * (q + e)) return;

  const int64_t T16 = (int64_t)tblk * 16;
  uint8_t *arena = (uint8_t *)calloc((size_t)(tblk * 6 + qblk + 1), 16);
  uint8_t *U = arena, *V = U + T16, *X = V + T16, *Y = X + T16, *S = Y + T16;
  uint8_t *SF = S + T16, *QR = SF + T16;
  int32_t *H = 0;
  if (!approx) {
    H = (int32_t *)malloc((size_t)T16 * sizeof(int32_t));
    for (int64_t k = 0; k < T16; ++k) H[k] = SDFO_NEG_INF;
  }
  const int64_t nrow = (int64_t)qlen + tlen - 1;
  const int64_t stride = (int64_t)ncol * 16;
  uint8_t *dir = 0;
  int32_t *row_lo = 0, *row_hi = 0;
  if (want_cigar) {
    dir = (uint8_t *)malloc((size_t)(nrow * stride + 16));
    row_lo = (int32_t *)malloc((size_t)nrow * 2 * sizeof(int32_t));
    row_hi = row_lo + nrow;
  }
  for (int k = 0; k < qlen; ++k) QR[k] = query[qlen - 1 - k];
  memcpy(SF, target, (size_t)tlen);

  int32_t H0 = 0;
  int last_H0_t = 0;
  int prev_lo = -1, prev_hi = -1;
  for (int r = 0; r < nrow; ++r) {
    int lo = 0, hi = tlen - 1;
    if (lo < r - qlen + 1) lo = r - qlen + 1;
    if (hi > r) hi = r;
    if (lo < ((r - w + 1) >> 1)) lo = (r - w + 1) >> 1;
    if (hi > ((r + w) >> 1)) hi = (r + w) >> 1;
    if (lo > hi) { ez->zdropped = 1; break; }
    const int lo0 = lo, hi0 = hi;
    lo = lo / 16 * 16;
    hi = (hi + 16) / 16 * 16 - 1;

    /* carry-in for cell `lo` (the (r-1, lo-1) neighbour) */
    int8_t cx, cv;
    if (lo > 0) {
      if (lo - 1 >= prev_lo && lo - 1 <= prev_hi) { cx = s8(X[lo - 1]); cv = s8(V[lo - 1]); }
      else cx = cv = 0;
    } else { cx = 0; cv = r ? q : 0; }
    if (hi >= r) { Y[r] = 0; U[r] = r ? q_b : 0; }

    /* score refresh */
    const uint8_t *qrow = QR + (qlen - 1 - r); /* qrow[t] = query[r - t] */
    if (!(flag & SDFO_GENERIC_SC)) {
      for (int t = lo0; t <= hi0; t += 16) {
        uint8_t a16[16], b16[16];
        memcpy(a16, SF + t, 16);
        memcpy(b16, qrow + t, 16);
        for (int k = 0; k < 16; ++k) {
          uint8_t sc = a16[k] == b16[k] ? sc_match : sc_mis;
          if (a16[k] == wild || b16[k] == wild) sc = 0;
          S[t + k] = sc; /* may spill into SF when t + k >= T16, exactly as the arena does */
        }
      }
    } else {
      for (int t = lo0; t <= hi0; ++t) S[t] = (uint8_t)mat[SF[t] * m + qrow[t]];
    }

    /* recurrence over the widened band; in-order over t so that the "old" (r-1) values of
       cell t-1 are carried in registers exactly like the byte shifts of the SIMD code */
    uint8_t carry_x = (uint8_t)cx, carry_v = (uint8_t)cv;
    /* sign-extension artefact: a negative carry byte also sets lanes 1..3 of the first block */
    const uint8_t smear_x = cx < 0 ? 0xff : 0, smear_v = cv < 0 ? 0xff : 0;
    uint8_t *drow = want_cigar ? dir + (int64_t)r * stride : 0;
    if (want_cigar) { row_lo[r] = lo; row_hi[r] = hi; }
    const int right = !!(flag & SDFO_RIGHT);
    for (int t = lo; t <= hi; ++t) {
      uint8_t xo = X[t], vo = V[t], uo = U[t];
      uint8_t xt1 = carry_x, vt1 = carry_v;
      if (t - lo >= 1 && t - lo <= 3) { xt1 |= smear_x; vt1 |= smear_v; }
      carry_x = xo; carry_v = vo;
      uint8_t z = (uint8_t)(S[t] + qe2_b);
      uint8_t a = (uint8_t)(xt1 + vt1);
      uint8_t b = (uint8_t)(Y[t] + uo);
      uint8_t d;
      if (!right) {
        d = s8(a) > s8(z) ? 1 : 0;
        z = smax8(z, a);
        if (s8(b) > s8(z)) d = 2;
      } else {
        d = s8(z) > s8(a) ? 0 : 1;
        z = smax8(z, a);
        if (!(s8(z) > s8(b))) d = 2;
      }
      if (b > z) z = b;         /* unsigned max */
      if (z > cap_b) z = cap_b; /* unsigned min */
      U[t] = (uint8_t)(z - vt1);
      V[t] = (uint8_t)(z - uo);
      z = (uint8_t)(z - q_b);
      a = (uint8_t)(a - z);
      b = (uint8_t)(b - z);
      if (!right) {
        if (s8(a) > 0) { X[t] = a; d |= 0x08; } else X[t] = 0;
        if (s8(b) > 0) { Y[t] = b; d |= 0x10; } else Y[t] = 0;
      } else {
        if (!(0 > s8(a))) { X[t] = a; d |= 0x08; } else X[t] = 0;
        if (!(0 > s8(b))) { Y[t] = b; d |= 0x10; } else Y[t] = 0;
      }
      if (want_cigar) drow[t - lo] = d;
    }

    if (!approx) {
      int32_t best, best_t;
      if (r > 0) {
        best = H[hi0] = hi0 > 0 ? H[hi0 - 1] + U[hi0] - qe : H[hi0] + V[hi0] - qe;
        best_t = hi0;
        int32_t lane_best[4], lane_t[4];
        for (int k = 0; k < 4; ++k) { lane_best[k] = best; lane_t[k] = best_t; }
        const int vec_end = lo0 + (hi0 - lo0) / 4 * 4;
        int t = lo0;
        for (; t < vec_end; t += 4)
          for (int k = 0; k < 4; ++k) {
            H[t + k] += (int32_t)V[t + k] - qe;
            if (H[t + k] > lane_best[k]) { lane_best[k] = H[t + k]; lane_t[k] = t; }
          }
        for (int k = 0; k < 4; ++k)
          if (best < lane_best[k]) { best = lane_best[k]; best_t = lane_t[k] + k; }
        for (; t < hi0; ++t) {
          H[t] += (int32_t)V[t] - qe;
          if (H[t] > best) { best = H[t]; best_t = t; }
        }
      } else {
        H[0] = (int32_t)V[0] - qe - qe;
        best = H[0]; best_t = 0;
      }
      if (hi0 == tlen - 1 && H[hi0] > ez->mte) { ez->mte = H[hi0]; ez->mte_q = r - hi; }
      if (r - lo0 == qlen - 1 && H[lo0] > ez->mqe) { ez->mqe = H[lo0]; ez->mqe_t = lo0; }
      if (track_max(ez, best, r, best_t, zdrop, e)) break;
      if (r == qlen + tlen - 2 && hi0 == tlen - 1) ez->score = H[tlen - 1];
    } else {
      if (r > 0) {
        if (last_H0_t >= lo0 && last_H0_t <= hi0 && last_H0_t + 1 >= lo0 && last_H0_t + 1 <= hi0) {
          int32_t d0 = (int32_t)V[last_H0_t] - qe;
          int32_t d1 = (int32_t)U[last_H0_t + 1] - qe;
          if (d0 > d1) H0 += d0; else { H0 += d1; ++last_H0_t; }
        } else if (last_H0_t >= lo0 && last_H0_t <= hi0) {
          H0 += (int32_t)V[last_H0_t] - qe;
        } else {
          ++last_H0_t;
          H0 += (int32_t)U[last_H0_t] - qe;
        }
        if ((flag & SDFO_APPROX_DROP) && track_max(ez, H0, r, last_H0_t, zdrop, e)) break;
      } else { H0 = (int32_t)V[0] - qe - qe; last_H0_t = 0; }
      if (r == qlen + tlen - 2 && hi0 == tlen - 1) ez->score = H0;
    }
    prev_lo = lo; prev_hi = hi;
  }
  free(arena);
  free(H);
  if (want_cigar) {
    cigbuf cg = {0, 0, 0};
    int rev = !!(flag & SDFO_REV_CIGAR);
    if (!ez->zdropped && !(flag & SDFO_EXTZ_ONLY))
      traceback(dir, row_lo, row_hi, stride, tlen - 1, qlen - 1, rev, &cg);
    else if (ez->max_t >= 0 && ez->max_q >= 0)
      traceback(dir, row_lo, row_hi, stride, ez->max_t, ez->max_q, rev, &cg);
    ez->cigar = cg.v;
    ez->n_cigar = cg.n;
    free(dir);
    free(row_lo);
  }
}

int64_t sdfo_band_cells(int qlen, int tlen, int w) {
  if (qlen <= 0 || tlen <= 0) return 0;
  if (w < 0) w = tlen > qlen ? tlen : qlen;
  int64_t cells = 0;
  for (int r = 0; r < qlen + tlen - 1; ++r) {
    int lo = 0, hi = tlen - 1;
    if (lo < r - qlen + 1) lo = r - qlen + 1;
    if (hi > r) hi = r;
    if (lo < ((r - w + 1) >> 1)) lo = (r - w + 1) >> 1;
    if (hi > ((r + w) >> 1)) hi = (r + w) >> 1;
    if (lo > hi) break;
    cells += hi - lo + 1;
  }
  return cells;
}

int64_t sdfo_extz2_batch(int64_t n, const uint8_t *pool, const int64_t *q_off,
                         const int32_t *qlen, const int64_t *t_off, const int32_t *tlen, int m,
                         const int8_t *mat, int gapo, int gape, int w, int zdrop, int flag,
                         int32_t *score_out, uint64_t *cigar_hash_out) {
  int64_t cells = 0;
  for (int64_t k = 0; k < n; ++k) {
    sdfo_result ez;
    sdfo_extz2(qlen[k], pool + q_off[k], tlen[k], pool + t_off[k], m, mat, gapo, gape, w, zdrop,
               flag, &ez);
    if (score_out) score_out[k] = ez.score;
    if (cigar_hash_out) {
      uint64_t h = 1469598103934665603ull; /* FNV-1a over the CIGAR words */
      for (int64_t c = 0; c < ez.n_cigar; ++c) {
        h ^= ez.cigar[c];
        h *= 1099511628211ull;
      }
      cigar_hash_out[k] = h;
    }
    free(ez.cigar);
    cells += sdfo_band_cells(qlen[k], tlen[k], w);
  }
  return cells;
}

void sdfo_cigar_counts(const uint32_t *cigar, int64_t n_cigar, const uint8_t *query,
                       const uint8_t *target, int32_t *matches, int32_t *mismatches,
                       int32_t *gaps, int32_t *gap_bases) {
  int64_t i = 0, j = 0; /* i: target, j: query */
  int32_t ma = 0, mi = 0, g = 0, gb = 0;
  for (int64_t c = 0; c < n_cigar; ++c) {
    uint32_t op = cigar[c] & 0xfu;
    int64_t len = cigar[c] >> 4;
    if (op == 0) {
      for (int64_t k = 0; k < len; ++k, ++i, ++j) {
        if (query[j] < 4 && target[i] < 4 && query[j] == target[i]) ++ma; else ++mi;
      }
    } else {
      ++g;
      gb += (int32_t)len;
      if (op == 1) j += len; else i += len;
    }
  }
  *matches = ma; *mismatches = mi; *gaps = g; *gap_bases = gb;
}
