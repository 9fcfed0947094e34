// TEST INFRASTRUCTURE: batch driver around the REFERENCE kernel for bench.py's cpu_baseline leg.
// Compiled together with the reference's own extern/ksw2_extz2_sse.cc (see Makefile `ref`);
// it only loops over tasks and calls ksw_extz2_sse (reference: extern/ksw2.h:50) the way
// align_helper does (reference: src/align.cc:49-66), freeing each CIGAR.
#include <assert.h>  // ksw2.h uses assert() without including it
#include <stdint.h>
#include <stdlib.h>

#include "ksw2.h"

extern "C" int64_t ref_extz2_batch(int64_t n, const uint8_t *pool, const int64_t *q_off,
                                   const int32_t *qlen, const int64_t *t_off, const int32_t *tlen,
                                   int m, const int8_t *mat, int gapo, int gape, int w, int zdrop,
                                   int flag, int32_t *score_out, uint64_t *cigar_hash_out) {
  int64_t done = 0;
  for (int64_t k = 0; k < n; ++k) {
    ksw_extz_t ez;
    ksw_extz2_sse(0, qlen[k], pool + q_off[k], tlen[k], pool + t_off[k], (int8_t)m, mat,
                  (int8_t)gapo, (int8_t)gape, w, zdrop, flag, &ez);
    if (score_out) score_out[k] = ez.score;
    if (cigar_hash_out) {
      uint64_t h = 1469598103934665603ull;
      for (int64_t c = 0; c < ez.n_cigar; ++c) {
        h ^= ez.cigar[c];
        h *= 1099511628211ull;
      }
      cigar_hash_out[k] = h;
    }
    free(ez.cigar);
    ++done;
  }
  return done;
}

// Single-task entry with the oracle's signature (oracle/extz2_oracle.h: sdfo_extz2) around the REFERENCE kernel, so
// that the host pipeline's test hook can run the stage-scale CPU leg on ksw_extz2_sse itself.
struct hook_result {  // layout of sdfo_result
  uint32_t max;
  int32_t zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score;
  int64_t n_cigar;
  uint32_t *cigar;
};
extern "C" void ref_extz2_hook(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                               const int8_t *mat, int gapo, int gape, int w, int zdrop, int flag, hook_result *out) {
  ksw_extz_t ez;
  ksw_extz2_sse(0, qlen, query, tlen, target, (int8_t)m, mat, (int8_t)gapo, (int8_t)gape, w, zdrop, flag, &ez);
  out->max = ez.max;
  out->zdropped = ez.zdropped;
  out->max_q = ez.max_q;
  out->max_t = ez.max_t;
  out->mqe = ez.mqe;
  out->mqe_t = ez.mqe_t;
  out->mte = ez.mte;
  out->mte_q = ez.mte_q;
  out->score = ez.score;
  out->n_cigar = ez.n_cigar;
  out->cigar = ez.cigar;  // malloc'd by the reference, the caller free()s (src/align.cc:65)
}
