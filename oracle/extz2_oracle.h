/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the `sedef align` DP hot path.
 *
 * This header and extz2_oracle.c are a plain-C restatement of the algorithm of the
 * reference's vendored ksw2 kernel (reference: extern/ksw2_extz2_sse.cc:23-298,
 * extern/ksw2.h:98-177).  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and only as the
 * checker.  The shipped path (sedef_amd/csrc) never links or calls this code.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit (every result field and
 * every CIGAR word) against (a) the golden vectors in tests/golden/ that were produced by
 * the reference kernel compiled from /root/reference (oracle/Makefile -> oracle/_ref/), and
 * (b) when oracle/_ref/libksw2_ref.so is present, live fuzzing against it
 * (tests/test_oracle_vs_ref.py).
 */
#ifndef SDF_EXTZ2_ORACLE_H
#define SDF_EXTZ2_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFO_NEG_INF (-0x40000000)

/* flag bits, numerically identical to the reference's KSW_EZ_* (extern/ksw2.h:8-16) */
#define SDFO_SCORE_ONLY 0x01
#define SDFO_RIGHT 0x02
#define SDFO_GENERIC_SC 0x04
#define SDFO_APPROX_MAX 0x08
#define SDFO_APPROX_DROP 0x10
#define SDFO_EXTZ_ONLY 0x40
#define SDFO_REV_CIGAR 0x80

/* mirrors ksw_extz_t (extern/ksw2.h:22-30) field by field */
typedef struct {
  uint32_t max;
  int32_t zdropped;
  int32_t max_q, max_t;
  int32_t mqe, mqe_t;
  int32_t mte, mte_q;
  int32_t score;
  int64_t n_cigar;
  uint32_t *cigar; /* malloc'd, len<<4|op, op 0=M 1=I 2=D; caller frees */
} sdfo_result;

/* One DP task; same argument meaning as ksw_extz2_sse (extern/ksw2.h:50). */
void sdfo_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                const int8_t *mat, int gapo, int gape, int w, int zdrop, int flag,
                sdfo_result *out);

/* Number of in-band DP cells of a task: sum over anti-diagonals of (en0-st0+1)
 * (extern/ksw2_extz2_sse.cc:101-114); the unit of the Gcell/s metric. */
int64_t sdfo_band_cells(int qlen, int tlen, int w);

/* Batch driver used by bench.py's cpu_baseline leg: tasks laid out in a code pool.
 * Returns total in-band cells; writes per-task score and a CIGAR checksum. */
int64_t sdfo_extz2_batch(int64_t n, const uint8_t *pool, const int64_t *q_off,
                         const int32_t *qlen, const int64_t *t_off, const int32_t *tlen, int m,
                         const int8_t *mat, int gapo, int gape, int w, int zdrop, int flag,
                         int32_t *score_out, uint64_t *cigar_hash_out);

/* Alignment column statistics of a CIGAR over code sequences (codes 0..3 = ACGT, >=4 = N):
 * restates populate_nice_alignment's counters (reference: src/align.cc:274-315) for
 * ACGTN-only input.  op 0=M consumes both, 1=I consumes query, 2=D consumes target. */
void sdfo_cigar_counts(const uint32_t *cigar, int64_t n_cigar, const uint8_t *query,
                       const uint8_t *target, int32_t *matches, int32_t *mismatches,
                       int32_t *gaps, int32_t *gap_bases);

#ifdef __cplusplus
}
#endif
#endif
