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
