/*
 * sedef_hip.h -- C ABI of the MI355X-native `sedef align` DP hot path.
 *
 * This is the drop-in boundary.  The reference has exactly one native interface on this path:
 *
 *     void ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen,
 *                        const uint8_t *target, int8_t m, const int8_t *mat, int8_t q, int8_t e,
 *                        int w, int zdrop, int flag, ksw_extz_t *ez);
 *                                              (reference: extern/ksw2.h:50, result ksw2.h:22-30)
 *
 * called once per DP task from align_helper (reference: src/align.cc:39-68).  A GPU needs the
 * tasks in batches, so the boundary is exported twice:
 *
 *   sdf_ksw_extz2()         same signature and result struct as ksw_extz2_sse -> 1-task drop-in;
 *   sdf_extz2_batch()       n tasks over a shared code pool; host buffers in, host buffers out;
 *   sdf_extz2_batch_device()  the same with inputs/outputs resident in HBM (no PCIe in the call).
 *
 * All entry points are plain C: pointers and sizes only.  Buffers are owned by the caller.
 * Every function that can fail returns 0 on success or a negative SDF_ERR_* code; the message
 * is available from sdf_last_error().  There is no CPU fallback: without a usable HIP device
 * sdf_create() fails.
 */
#ifndef SEDEF_HIP_H
#define SEDEF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDF_NEG_INF (-0x40000000) /* KSW_NEG_INF, reference extern/ksw2.h:6 */

/* task flag bits: numerically the reference's KSW_EZ_* (extern/ksw2.h:8-16) */
#define SDF_FLAG_SCORE_ONLY 0x01 /* no direction matrix, no CIGAR */
#define SDF_FLAG_RIGHT 0x02      /* right-align gaps */
#define SDF_FLAG_GENERIC_SC 0x04 /* scores from the whole m x m matrix (general kernel; SEDEF never sets it) */
#define SDF_FLAG_APPROX_MAX 0x08 /* approximate max: score only along one followed path (general kernel) */
#define SDF_FLAG_APPROX_DROP 0x10 /* with APPROX_MAX: z-drop test on the followed value */
#define SDF_FLAG_EXTZ_ONLY 0x40  /* traceback from the best extension cell */
#define SDF_FLAG_REV_CIGAR 0x80  /* leave the CIGAR reversed */

/* error codes */
#define SDF_OK 0
#define SDF_ERR_NO_DEVICE (-1)
#define SDF_ERR_HIP (-2)
#define SDF_ERR_UNSUPPORTED (-3) /* alphabet (m != 5) or flag bits the GPU path does not implement */
#define SDF_ERR_INVALID (-4)
#define SDF_ERR_CIGAR_OVERFLOW (-5) /* cigar_cap too small; *cigar_used holds the need */
#define SDF_ERR_NOMEM (-6)

/* what to compute for a batch (bit mask) */
#define SDF_WANT_CIGAR 0x1  /* CIGAR + n_cigar + column counts */
#define SDF_WANT_SCORE 0x2  /* score, mte, mte_q, zdropped (always produced) */
#define SDF_WANT_EXT 0x4    /* also max, max_q, max_t, mqe, mqe_t: every ksw_extz_t field.  Needed
                               for zdrop >= 0 and SDF_FLAG_EXTZ_ONLY; SEDEF reads none of them
                               (reference: src/align.cc:49-66 uses ez.cigar / ez.n_cigar only). */
#define SDF_WANT_ALL 0x7

typedef struct sdf_ctx sdf_ctx;

/* Scoring, as passed to ksw_extz2_sse: alphabet size m (must be 5: ACGT + wildcard), the m*m
 * matrix of which the non-generic kernel uses mat[0] (match), mat[1] (mismatch) and the
 * min over all entries (reference: extern/ksw2_extz2_sse.cc:66-67,77-81), gap open q, extend e. */
typedef struct {
  int32_t m;
  int8_t mat[25];
  int8_t gapo, gape;
  int8_t pad_;
} sdf_scoring;

/* One DP task.  q_off/t_off index the sequence pool: bytes for sdf_extz2_batch (one code 0..4
 * per byte, exactly what ksw_extz2_sse takes), 32-bit words of the packed pool for
 * sdf_extz2_batch_device (see sdf_pack_codes). */
typedef struct {
  int64_t q_off, t_off;
  int32_t qlen, tlen;
  int32_t w;     /* band width, <0 = full (reference default, src/align.cc:86) */
  int32_t zdrop; /* <0 = off */
  int32_t flag;
  int32_t pad_;
} sdf_task;

/* Per-task result record (64 B).  First nine fields mirror ksw_extz_t (extern/ksw2.h:22-30). */
typedef struct {
  int32_t score;
  int32_t max, max_q, max_t;
  int32_t mqe, mqe_t;
  int32_t mte, mte_q;
  int32_t zdropped;
  int32_t n_cigar;    /* number of CIGAR words */
  int64_t cigar_off;  /* first word in the CIGAR pool; words are len<<4|op, op 0=M 1=I 2=D */
  /* alignment column statistics over the CIGAR (the counters of populate_nice_alignment,
   * reference: src/align.cc:274-315, for ACGTN input): */
  int32_t matches, mismatches, gaps, gap_bases;
} sdf_result;

/* ksw_extz_t, field-for-field (reference: extern/ksw2.h:22-30) */
typedef struct {
  uint32_t max : 31, zdropped : 1;
  int max_q, max_t;
  int mqe, mqe_t;
  int mte, mte_q;
  int score;
  uint32_t *cigar; /* malloc'd by the callee, caller free()s (reference: src/align.cc:65) */
  int64_t m_cigar, n_cigar;
} sdf_ksw_extz_t;

/* ---- context ---------------------------------------------------------------------------- */
int sdf_device_count(void);
/* device: HIP ordinal.  workspace_bytes: HBM budget for direction matrices per internal
 * batch (0 = half of the memory free when the context is made; any figure is clamped to that).  The
 * workspace is allocated by need, up to the budget: a batch that needs more runs in chunks.
 * Returns NULL on failure (sdf_last_error(NULL) has the reason). */
sdf_ctx *sdf_create(int device, size_t workspace_bytes);
void sdf_destroy(sdf_ctx *ctx);

/* ---- configuration -------------------------------------------------------------------------------------------------
 * Every tunable and test switch of the library in ONE struct.  sdf_create() fills it from the SDF_* environment variables,
 * once, when the context is made (an unknown value or one out of range makes sdf_create fail with the reason);
 * sdf_create_cfg() takes it from the caller instead -- tests force a kernel through sdf_config_set(&cfg, "SDF_NO_PAIR",
 * "1", ...) and never touch the environment.  A context's settings do not change after it is made; the contexts it
 * creates for itself (the first part of a split batch, the re-run of tasks a stripe kernel gave up) inherit them.
 * Fields are int64_t (0 / 1 for switches) except workspace_gib; sdf_config_describe() lists name, environment
 * variable, default and meaning of each, sdf_config_dump() the values of one struct (SDF_DEBUG_PLAN=1 prints that for
 * every context made). */
typedef struct sdf_config {
  uint32_t size; /* sizeof(sdf_config), set by sdf_config_default / sdf_config_from_env */
  uint32_t reserved;
  /* which kernel serves a task */
  int64_t force_general, no_pair, no_mixed, mixed_min, self_pair_max;
  int64_t no_stripe, stripe_min, stripe_nreg, stripe_claim, stripe_spin_cap;
  int64_t bstripe_min_rows, bstripe_nreg, bstripe_all;
  int64_t no_strip, strip_always, strip_cols, chain_min;
  int64_t no_lane, lane_min, lane_plan_sort, lane_prio;
  /* how a batch is cut and planned */
  int64_t pipeline, cut_chunks, heavy_bytes, early_heavy, split_min, split_div;
  int64_t plan_threads, plan_pool_from, scan_pool_from, pool_spin_us;
  int64_t pin_register;
  double workspace_gib;
  /* other entry points */
  int64_t chain_threads_only, stats_items, stats_group_max;
  /* diagnostics on stderr */
  int64_t debug_plan, debug_timing, debug_classes, debug_plan_early;
} sdf_config;
void sdf_config_default(sdf_config *cfg);
/* defaults, then every SDF_* variable that is set; SDF_ERR_INVALID + message for a value that is not a number or out of range */
int sdf_config_from_env(sdf_config *cfg, char *err, size_t errcap);
/* one setting by its environment variable's or its field's name */
int sdf_config_set(sdf_config *cfg, const char *name, const char *value, char *err, size_t errcap);
size_t sdf_config_dump(const sdf_config *cfg, char *buf, size_t cap); /* "NAME=value" lines; returns the bytes needed */
size_t sdf_config_describe(char *buf, size_t cap);                    /* every setting: variable, field, default, meaning */
/* sdf_create with the caller's settings (cfg == NULL: sdf_config_from_env, which is what sdf_create does) */
sdf_ctx *sdf_create_cfg(int device, size_t workspace_bytes, const sdf_config *cfg);
const sdf_config *sdf_get_config(const sdf_ctx *ctx);
/* Sizes the context's buffers ONCE for host-buffer batch calls of up to max_tasks tasks whose sequences add up to
 * max_bases bases: the pinned staging (task plan, launch order, packed sequences, results), the device copies of
 * those, the CIGAR staging and the streams of the call's pipeline -- no call within the bounds allocates, pins or sets
 * up a hardware queue afterwards (pinning runs at ~4 MB per
 * millisecond: a first call of 700,000 tasks spends 40-50 ms on it) -- and workspace_bytes of the direction-flag
 * workspace (clamped to the context's budget; 0: left to the first call that needs it).  Larger calls still grow the
 * buffers.  Call it where a context is set up: pinning takes the process's memory-map lock, and a pageable upload that
 * runs meanwhile (sdf_anchors_batch) waits for it -- measured: 46-78 ms for a 34-54 MB upload next to a reserve on another
 * thread, 4 ms without.  It must have returned before the context's first batch call.  (The stage driver: once per lane,
 * with the lane's context, host/pipeline.cc.) */
#define SDF_RESERVE_BRIEF 1u /* the caller reads results through sdf_extz2_batch_brief: 16 bytes of result staging per task */
#define SDF_RESERVE_ANCHORS 2u /* the caller will use sdf_anchors_batch: one tiny call now, so that the first real one does not
                                  pay for the first launch of its kernels and of the library sort behind it */
#define SDF_RESERVE_FEW_STREAMS 4u /* this context is one of several that share the device (the lanes of the stage driver): it
                                       never creates the four extra pipeline streams -- a stream costs 7-15 ms to set up, and the
                                       other contexts' work fills the device where the extra streams would */
int sdf_reserve(sdf_ctx *ctx, size_t max_tasks, size_t max_bases, size_t workspace_bytes, uint32_t flags);
/* Debug: out[4096] receives the wavefronts the chained-strip launches started per (XCD, shader engine, CU, SIMD) since the
 * last call (index xcd << 9 | se << 6 | cu << 2 | simd) and the counters are cleared; the first call (out may be NULL) switches
 * the counting on for the process. */
int sdf_debug_placement(sdf_ctx *ctx, uint32_t *out);
/* Device bytes the context holds at this moment (buffers in use, outgrown ones not yet freed). */
size_t sdf_device_bytes(const sdf_ctx *ctx);
const char *sdf_last_error(const sdf_ctx *ctx);

/* ---- packed sequence format --------------------------------------------------------------
 * A sequence of len codes occupies sdf_packed_words(len) 32-bit words:
 *   ceil(len/16) words of 2-bit codes (base b in bits 2*(b%16) of word b/16; N stored as 0)
 *   ceil(len/32) words of N mask      (bit b%32 of word b/32 set <=> code >= 4). */
size_t sdf_packed_words(int32_t len);
void sdf_pack_codes(const uint8_t *codes, int32_t len, uint32_t *out);
/* Packs both sequences of n tasks (byte offsets into `codes`, as sdf_extz2_batch takes them) back to back into `out`
 * and writes each task's word offsets, as sdf_extz2_batch_device takes them.  `out` holds
 * sum(sdf_packed_words(qlen) + sdf_packed_words(tlen)) words.  Returns that number. */
size_t sdf_pack_tasks(const uint8_t *codes, const int64_t *q_off, const int32_t *qlen, const int64_t *t_off,
                      const int32_t *tlen, size_t n, uint32_t *out, int64_t *q_word, int64_t *t_word);

/* ---- batched DP ---------------------------------------------------------------------------
 * Host-buffer form.  seq_pool holds byte codes; out[n]; cigar_pool[cigar_cap] words receives
 * all CIGARs back to back in task order (out[i].cigar_off, out[i].n_cigar); *cigar_used gets
 * the number of words written (or needed, with SDF_ERR_CIGAR_OVERFLOW). */
int sdf_extz2_batch(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                    const uint8_t *seq_pool, size_t pool_bytes, uint32_t want, sdf_result *out,
                    uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);

/* The same call with 16-byte result records: what a caller that only stitches CIGARs reads (the stage driver:
 * src/align.cc:129-175 keeps the CIGAR of every piece and nothing else of ksw_extz_t) -- a quarter of the bytes
 * of sdf_result across PCIe and through the caller's caches.  Always computes CIGARs. */
typedef struct {
  int64_t cigar_off; /* first word of the task's CIGAR in cigar_pool */
  int32_t n_cigar;
  int32_t matches;   /* 'M' columns with equal bases (what src/align.cc:274-311 counts as error.matches) */
} sdf_result_brief;
int sdf_extz2_batch_brief(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                          const uint8_t *seq_pool, size_t pool_bytes, sdf_result_brief *out,
                          uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);


/* ---- resident sequences: DP tasks that name ranges of characters already in HBM ------------------------------------
 * The reference's align_helper reads the bases of every DP call in place: the caller hands it two substrings of the pair's
 * sequences, mapped through align_dna (reference: src/align.cc:39-57,80-84; src/common.h:60-70,91).  The stage driver has
 * each super-batch's FASTA characters in HBM already -- sdf_anchors_batch uploaded them for the seed anchors -- and its DP
 * rounds (hundreds of thousands of gap fills of ~25 bases) used to cut the same bases out again on the host, code them,
 * pack them and upload them, round after round.  Here a task's q_off / t_off are BYTE offsets into that resident pool of raw
 * FASTA characters; the device does align_dna (ACGT of either case -> 0..3, anything else -> the wildcard 4) and the
 * packing (seq_pack.hip).  Results as sdf_extz2_batch_brief / sdf_extz2_batch return them.
 *
 *   sdf_pool_host(ctx, bytes)       a pinned host buffer of at least `bytes` owned by the context (NULL on failure): filled
 *                                   there, a pool crosses PCIe as one asynchronous DMA instead of a staged pageable copy
 *                                   (182 MB: 26 ms pageable).  Valid until the next sdf_pool_host call or sdf_destroy.
 *   sdf_pool_upload(ctx, p, bytes)  copies `bytes` characters to HBM, enqueued on the context's stream (any host memory;
 *                                   the buffer must stay unchanged until the next call on this context that returns data).
 *                                   The pool replaces the one before.
 *   sdf_anchors_batch(..., seq_pool = NULL, pool_bytes, ...)   the seed anchors of pairs whose offsets point into the
 *                                   resident pool; with a host seq_pool the call uploads it and leaves it resident.
 *   sdf_extz2_batch_pairs           the DP batch on ranges of the resident pool, 16-byte results;
 *   sdf_extz2_batch_pairs_full      the same with sdf_result records and a `want` mask.
 *   sdf_extz2_batch_pairs_view / sdf_anchors_batch_view   the same calls for a caller that reads the results where the device's
 *                                   copies land -- the context's pinned staging -- instead of receiving a second copy in
 *                                   arrays of its own (the stage driver: 17 MB of records and CIGAR words per round, 34 MB of
 *                                   anchors per super-batch).  *out / *cigar_pool are valid until the context's next call of
 *                                   the same kind. */
char *sdf_pool_host(sdf_ctx *ctx, size_t bytes);
int sdf_pool_upload(sdf_ctx *ctx, const char *chars, size_t bytes);
size_t sdf_pool_bytes(const sdf_ctx *ctx); /* characters resident at this moment */
int sdf_extz2_batch_pairs(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, sdf_result_brief *out,
                          uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);
int sdf_extz2_batch_pairs_full(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, uint32_t want,
                               sdf_result *out, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used);
int sdf_extz2_batch_pairs_view(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                               const sdf_result_brief **out, const uint32_t **cigar_pool, size_t *cigar_used);

/* Device-resident form: d_packed_pool, d_out and d_cigar_pool are HBM pointers on ctx's device;
 * tasks (host) carry word offsets into d_packed_pool.  Work is enqueued on `stream`
 * (a hipStream_t, NULL = the context's own stream) and the call returns after the stream has
 * drained (it needs *cigar_used).  */
int sdf_extz2_batch_device(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                           const uint32_t *d_packed_pool, uint32_t want, sdf_result *d_out,
                           uint32_t *d_cigar_pool, size_t cigar_cap, size_t *cigar_used,
                           void *stream);

/* In-band DP cells of a task: the unit of the Gcell/s metric
 * (sum over anti-diagonals of en0-st0+1, reference: extern/ksw2_extz2_sse.cc:101-114). */
int64_t sdf_band_cells(int32_t qlen, int32_t tlen, int32_t w);

/* Timing of the last batch call, from HIP events recorded on the launch streams:
 * which = 0 DP kernels, 1 traceback, 2 CIGAR compaction, 3 whole call (stream time);
 * 4 host-side planning before the first launch, 5 whole call on the host clock.
 * Large batches run as a pipeline of chunks (planning of chunk i+1, DP of chunk i+1 and traceback of chunk i
 * overlap); 0 and 1 are then the time during which at least one DP / traceback kernel was in flight and
 * 6 is the sum of the chunks' DP intervals.  SDF_PIPELINE=0 in the environment of sdf_create() keeps a batch
 * in one chunk on one stream. */
float sdf_last_ms(const sdf_ctx *ctx, int which);
/* Number of DP kernel launches in the last batch call and algorithmic bytes they moved. */
int sdf_last_launches(const sdf_ctx *ctx);
/* Number of tasks of the last batch call that ran two per wavefront: tasks with the same (qlen, tlen, w, flag)
 * share the reference's band schedule (extern/ksw2_extz2_sse.cc:101-115) and are packed side by side. */
long long sdf_last_paired(const sdf_ctx *ctx);
/* Number of tasks of the last batch call that were run a second time, inside the call, on the one-wavefront / one-workgroup
 * kernels because a stripe kernel's wavefront gave up waiting for its neighbour (SDF_STRIPE_SPIN_CAP polls; the stripe
 * protocol's forward progress rests on the dispatch order).  0 in normal operation. */
long long sdf_last_reran(const sdf_ctx *ctx);
/* Number of tasks of the last batch call that ran one per LANE (extz2_lane.hip): small full-band tasks -- both sequences of
 * at most 256 bases, at most 16,384 cells; SEDEF's gap fills (reference: src/align.cc:235) -- of a batch that holds at
 * least 8,192 of them, under a scoring with match + 2 (gap open + gap extend) <= 127.  They are sorted and planned on the
 * device; the host only marks them. */
long long sdf_last_lane_tasks(const sdf_ctx *ctx);

/* ---- seed anchors on the GPU (next row of the scope table) -----------------------------------
 * Replaces generate_anchors (reference: src/chain.cc:24-101) for a batch of candidate pairs: maximal exact
 * k-mer matches, in the reference's order (query position, then reference position).  Sequences are the raw
 * FASTA characters (case = soft-masking, N = unknown).  kmer <= 15 and sequences shorter than 2 Gb (SDF_ERR_UNSUPPORTED
 * otherwise); any number of pairs per call (the call runs them in ranges that fit its 64-bit sort key).  seq_pool = NULL: the
 * offsets point into the pool sdf_pool_upload left in HBM (pool_bytes: how much of it the pairs may name). */
typedef struct {
  int64_t q_off, r_off; /* byte offsets of query / reference characters in seq_pool */
  int32_t qlen, rlen;
  int32_t same_chr;     /* 1: same chromosome and strand -> k-mers within `kmer` of the main diagonal are skipped */
  int32_t delta;        /* ref_start - query_start of the pair (src/chain.cc:67-69) */
} sdf_anchor_pair;

typedef struct {
  int32_t q, r, l, has_u; /* reference: struct Anchor, src/align.h:25-28 */
} sdf_anchor;

/* out[out_off[i] .. out_off[i+1]) are the anchors of pair i (out_off has n+1 entries).  *out_used receives the
 * total; with SDF_ERR_CIGAR_OVERFLOW it holds the capacity needed. */
int sdf_anchors_batch(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *seq_pool, size_t pool_bytes,
                      int kmer, sdf_anchor *out, size_t out_cap, int64_t *out_off, size_t *out_used);
/* ... with the anchors left in the context's pinned staging (*out: valid until the context's next anchors call) */
int sdf_anchors_batch_view(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *seq_pool, size_t pool_bytes,
                           int kmer, const sdf_anchor **out, int64_t *out_off, size_t *out_used);
/* ... of more pairs of the resident pool, written BEHIND the first `keep` anchors of that staging, which stay valid (the stage
 * driver chains one half of a super-batch while the device finds the anchors of the other).  out_off counts from *out.  The
 * staging does not grow here: SDF_ERR_CIGAR_OVERFLOW when it has no room (*out_used: the anchors there would be). */
int sdf_anchors_batch_more(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, size_t pool_bytes, int kmer, size_t keep,
                           const sdf_anchor **out, int64_t *out_off, size_t *out_used);

/* ---- anchor chaining on the GPU ---------------------------------------------------------------
 * Replaces chain_anchors (reference: src/chain.cc:103-199) for a batch of pairs whose anchors are laid out as
 * sdf_anchors_batch returns them: anchors[off[i] .. off[i+1]).  The reference returns, per pair, `path` (anchor
 * indices, best chain first, each chain from its last anchor backwards) and `boundaries` ({end position in path,
 * any-uppercase flag}, starting with {0, 0}):
 *   path[off[i] + k]                    k-th path element of pair i (index within the pair), k < off[i+1]-off[i]
 *   bounds[2 * (off[i] + i + b) + 0/1]  b-th boundary of pair i, b < nbound[i]
 * path holds off[n] entries, bounds 2 * (off[n] + n), nbound n.  max_chain_gap / match_chain_score are the
 * reference's MAX_CHAIN_GAP / MATCH_CHAIN_SCORE (src/common.h). */
int sdf_chain_batch(sdf_ctx *ctx, const sdf_anchor *anchors, const int64_t *off, size_t n, int max_chain_gap,
                    int match_chain_score, int32_t *path, int32_t *bounds, int32_t *nbound);

/* ---- per-alignment columns of `stats generate` on the GPU (scope row f4) -------------------------
 * Replaces the column walk of process() (reference: src/stats_main.cc:228-270) and the AlignmentError counters of
 * populate_nice_alignment (src/align.cc:300-314) for a batch of finished alignments.  An alignment is the two
 * sequences as FASTA characters (case = soft-masking) and its CIGAR as runs `len << 4 | op` with op 0 = 'M',
 * 1 = 'D' (consumes a only), 2 = 'I' (consumes b only) -- the reference's letters (src/align.cc:59-63), numerically
 * the words sdf_extz2_batch returns.  Nothing is expanded into column strings.  Sequences up to 16 Mb
 * (SDF_ERR_UNSUPPORTED beyond); a CIGAR that consumes more than its sequences, where the reference would read past
 * its strings, sets `flags` to 1 and makes the host-buffer call return SDF_ERR_INVALID. */
typedef struct {
  uint64_t a_off, b_off;   /* byte offsets of the two sequences in seq_pool */
  uint32_t a_len, b_len;
  uint64_t cigar_off;      /* first run of the alignment in cigar_pool (in words) */
  uint32_t n_cigar, reserved;
} sdf_stats_task;

typedef struct {
  /* src/stats_main.cc:231-270, in the order of the output columns 15-21 and 27-29 */
  int32_t indel_a, indel_b, aln_b, match_b, mismatch_b, transitions_b, transversions_b;
  int32_t uppercase_a, uppercase_b, uppercase_matches;
  /* Alignment::matches() / mismatches() / gaps() / gap_bases() / span() (src/align.h:79-83) */
  int32_t matches, mismatches, gaps, gap_bases, span;
  int32_t flags;           /* 1: the CIGAR does not fit the sequences (counters undefined) */
} sdf_stats_cols;

int sdf_stats_columns_batch(sdf_ctx *ctx, const sdf_stats_task *tasks, size_t n, const char *seq_pool, size_t pool_bytes,
                            const uint32_t *cigar_pool, size_t cigar_words, sdf_stats_cols *out);
/* The same with tasks, pools and results resident in HBM; asynchronous on `stream` (a hipStream_t; NULL = the
 * context's own stream, synchronised before returning).  Offsets are not checked against the pools here.  One
 * wavefront takes one alignment; an alignment of more than 1,024 runs is cut into segments of 512 runs on the device, which
 * wavefronts of a second launch count side by side (a list of 2^18 segments in the context: calls on one context do not
 * overlap). */
int sdf_stats_columns_device(sdf_ctx *ctx, const sdf_stats_task *d_tasks, size_t n, const char *d_seq_pool,
                             const uint32_t *d_cigar_pool, sdf_stats_cols *d_out, void *stream);

/* ---- multi-GPU: the one exchange step of the path.  DP tasks are independent (the reference runs one
 * single-threaded process per bucket file and concatenates their output files, sedef.sh:187-190,218-221), so a batch is
 * sharded over the GPUs of a node with no data-path collective; after the DP an RCCL all-gatherv over xGMI gives every GPU
 * every shard's result records and CIGAR words.  One communicator per (process, device): either
 *   rank 0: sdf_comm_unique_id(id, 128) -> id handed to the other processes out of band -> every rank:
 *   sdf_comm_create(device, world, rank, id)                                   (one process per GPU), or
 *   sdf_comm_create_all(devices, n, comms)                                      (one process, a thread per GPU).
 * RCCL is dlopen'ed by the first of these calls. */
typedef struct sdf_comm sdf_comm;
int sdf_comm_unique_id(void *id, size_t bytes /* >= 128 */);
sdf_comm *sdf_comm_create(int device, int world, int rank, const void *id);
int sdf_comm_create_all(const int *devices, int n, sdf_comm **out);
void sdf_comm_destroy(sdf_comm *c);
int sdf_comm_world(const sdf_comm *c);
int sdf_comm_rank(const sdf_comm *c);
const char *sdf_comm_last_error(const sdf_comm *c);
/* All-gatherv of one batch's results: d_out[n_tasks] and d_cig[cig_used] are this rank's (HBM, what
 * sdf_extz2_batch_device left), d_all_out / d_all_cig receive every rank's back to back in rank order -- exactly
 * counts[2 r] records and counts[2 r + 1] CIGAR words from rank r (counts: host, 2 * world entries; cigar_off of a record
 * stays relative to its rank's words).  Two collectives whatever the world size: an all-gather of the counts and one
 * group of point-to-point transfers on the exact sizes.  Enqueued on `stream` (NULL: the communicator's own, synchronised
 * before returning); with a stream the call returns once the counts are on the host and the transfers are enqueued.
 * SDF_ERR_CIGAR_OVERFLOW: a capacity is too small ON ANY RANK (counts holds the sizes): the capacities travel with the
 * counts, so every rank returns this together and none is left waiting in a receive. */
int sdf_allgatherv_results(sdf_comm *c, const sdf_result *d_out, size_t n_tasks, const uint32_t *d_cig, size_t cig_used,
                           sdf_result *d_all_out, size_t all_out_cap, uint32_t *d_all_cig, size_t all_cig_cap,
                           uint64_t *counts, void *stream);

/* ---- one-task drop-in: same contract as ksw_extz2_sse (extern/ksw2.h:50).  `km` is ignored
 * like in the reference build (no HAVE_KALLOC).  Uses a process-wide context on device 0 (or
 * the device named by SDF_DEVICE).  On a fatal error prints to stderr and exits with 120, the
 * reference's own failure mode (extern/ksw2.h:106). */
void sdf_ksw_extz2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                   int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                   sdf_ksw_extz_t *ez);

#ifdef __cplusplus
}
#endif
#endif
