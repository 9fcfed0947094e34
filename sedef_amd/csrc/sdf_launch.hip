// Device side of a batch call: uploads a planned chunk, launches its DP kernels and its traceback, and closes the
// batch (CIGAR scan + compaction, timing).  Planning is in sdf_plan.hip; the entry points are in sdf_api.hip.
#include "sdf_ctx.h"

namespace sdf {

__global__ __launch_bounds__(256) void reset_results_kernel(sdf_result *res, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  sdf_result o;  // ksw_reset_extz (reference: extern/ksw2.h:153-159)
  o.score = o.mqe = o.mte = SDF_NEG_INF;
  o.max = 0;
  o.max_q = o.max_t = o.mqe_t = o.mte_q = -1;
  o.zdropped = 0;
  o.n_cigar = 0;
  o.cigar_off = 0;
  o.matches = o.mismatches = o.gaps = o.gap_bases = 0;
  res[k] = o;
}

struct ChunkEv {
  hipEvent_t dp0 = nullptr, dpe[16] = {}, tb0 = nullptr, tb1 = nullptr;  // plan uploaded; end of the DP launches per
                                                                         // stream; traceback (begin, end)
};

constexpr size_t kClaimSets = 2048;  // stripe launches per call that take their entries through counters (the rest by index)

// Progress of one batch call on the device.
struct BatchRun {
  sdf_ctx *ctx = nullptr;
  hipStream_t st = nullptr;  // the caller's stream
  ScoreK sk;
  const uint32_t *d_pool = nullptr;
  sdf_result *d_out = nullptr;
  PlanTask *plan = nullptr, *d_plan = nullptr;  // pinned host copy / device copy
  int32_t *order = nullptr, *d_order = nullptr;
  uint8_t *d_dir = nullptr;
  uint8_t *heavy_dir = nullptr;  // the heavy chunks' slice (the workspace as it was when they were launched)
  size_t claim_sets = 0;         // stripe launches of the call so far (each has eight entry counters in ctx->claim_buf)
  bool more_chunks = false;      // early start: chunks of ordinary tasks will follow those in cut->chunks
  uint32_t *d_stage = nullptr;
  const BatchCut *cut = nullptr;
  bool want_cigar = false, have_heavy = false;
  std::vector<ChunkEv> cev;
  std::vector<size_t> normal_ids;  // chunk indices of the ordinary chunks launched so far
  double qload[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // estimated DP work queued on each stream during this call
  size_t evc = 0;
  hipEvent_t ev_begin = nullptr;
  bool any_stripe = false;  // a stripe kernel was launched: its give-up list is looked at before the batch closes
  const sdf_scoring *scoring = nullptr;  // what the caller passed (a re-run of abandoned tasks passes them on)
  const sdf_task *tasks = nullptr;
  uint32_t want = 0;
  hipEvent_t ev_lane = nullptr;  // the lane tasks' DP and traceback have finished
  hipEvent_t ev_lane0 = nullptr;  // ... are about to start (debug timing)
  hipStream_t began[24] = {};  // internal streams already ordered behind ev_begin in this call
  size_t nbegan = 0;
};

static hipEvent_t next_event(sdf_ctx *ctx, size_t &cursor) {
  if (cursor == ctx->events.size()) {
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    ctx->events.push_back(ev);
  }
  return ctx->events[cursor++];
}

// every stream a batch call may have work on
static void drain_streams(sdf_ctx *ctx, hipStream_t st) {
  for (hipStream_t q : {st, ctx->stream, ctx->dp_stream[0], ctx->dp_stream[1], ctx->tb_stream, ctx->aux_stream[0],
                        ctx->aux_stream[1], ctx->aux_stream[2], ctx->aux_stream[3], ctx->lane_stream})
    if (q) (void)hipStreamSynchronize(q);
  for (hipStream_t q : ctx->wide_stream)
    if (q) (void)hipStreamSynchronize(q);
  (void)hipGetLastError();
}

// One DP launch of a planned class.  slabs: HBM state of the very long tasks (HBM-state classes only).
static void launch_dp(const Launch &L, hipStream_t sdp, const PlanTask *lp, const int32_t *lo, const uint32_t *d_pool,
                      const ScoreK &sk, uint8_t *dir_reg, sdf_result *d_out, uint8_t *slabs, unsigned long long *gave_up,
                      int spin_cap, unsigned *claim = nullptr) {
  const dim3 one((unsigned)L.cnt), half((unsigned)(L.cnt / 2));
#define SDF_WAVE(N, S) \
  hipLaunchKernelGGL((extz2_wave_kernel<N, S>), one, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out)
#define SDF_PAIR(N, S) \
  hipLaunchKernelGGL((extz2_pair_kernel<N, S, false>), half, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out)
#define SDF_PAIR_MIXED(N) \
  hipLaunchKernelGGL((extz2_pair_mixed_kernel<N>), half, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out)
#define SDF_PAIR_TRACK(N) \
  hipLaunchKernelGGL((extz2_pair_kernel<N, true, true>), half, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out)
#define SDF_STRIPE(N) /* one workgroup of one wavefront per stripe; progress words and edge columns reset first */ \
  {                                                                                                              \
    hipLaunchKernelGGL(stripe_sync_init_kernel, one, dim3(64), 0, sdp, lp, lo, N, dir_reg);                      \
    hipLaunchKernelGGL((extz2_stripe_kernel<N>), one, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out, L.rmax, gave_up, spin_cap, claim); \
  }
#define SDF_BSTRIPE(N) /* banded stripes: records and edge columns reset first, the records merged afterwards */    \
  {                                                                                                               \
    hipLaunchKernelGGL(bstripe_init_kernel, one, dim3(64), 0, sdp, lp, lo, N, dir_reg);                           \
    hipLaunchKernelGGL((extz2_bstripe_kernel<N>), one, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out, gave_up, spin_cap, claim); \
    hipLaunchKernelGGL(bstripe_finish_kernel, dim3((unsigned)((L.cnt + 63) / 64)), dim3(64), 0, sdp, lp, lo,     \
                       (int)L.cnt, N, dir_reg, d_out);                                                            \
  }
#define SDF_GENERAL(BS, PLAIN)                                                                                      \
  hipLaunchKernelGGL((extz2_general_kernel<BS, false, PLAIN>), one, dim3(BS), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, \
                     d_out, (uint8_t *)nullptr, (size_t)0)
#define SDF_GENERAL_HBM(BS, PLAIN) /* L.lds = per-workgroup slab bytes in HBM */                                    \
  hipLaunchKernelGGL((extz2_general_kernel<BS, true, PLAIN>), one, dim3(BS), 512, sdp, lp, lo, d_pool, sk, dir_reg, \
                     d_out, slabs, L.lds)
  switch (L.bs) {
    case 1: SDF_WAVE(1, false); break;
    case 11: SDF_WAVE(1, true); break;
    case 2: SDF_WAVE(2, false); break;
    case 12: SDF_WAVE(2, true); break;
    case 3: SDF_WAVE(3, false); break;
    case 13: SDF_WAVE(3, true); break;
    case 6: SDF_WAVE(6, false); break;
    case 16: SDF_WAVE(6, true); break;
    case 4: SDF_WAVE(4, false); break;
    case 14: SDF_WAVE(4, true); break;
    case 8: SDF_WAVE(8, false); break;
    case 18: SDF_WAVE(8, true); break;
    case 101: SDF_PAIR(1, false); break;
    case 111: SDF_PAIR(1, true); break;
    case 102: SDF_PAIR(2, false); break;
    case 112: SDF_PAIR(2, true); break;
    case 103: SDF_PAIR(3, false); break;
    case 113: SDF_PAIR(3, true); break;
    case 104: SDF_PAIR(4, false); break;
    case 114: SDF_PAIR(4, true); break;
    case 106: SDF_PAIR(6, false); break;
    case 116: SDF_PAIR(6, true); break;
    case 108: SDF_PAIR(8, false); break;
    case 118: SDF_PAIR(8, true); break;
    case 132: SDF_PAIR_MIXED(2); break;
    case 133: SDF_PAIR_MIXED(3); break;
    case 134: SDF_PAIR_MIXED(4); break;
    case 135: SDF_PAIR_MIXED(5); break;
    case 136: SDF_PAIR_MIXED(6); break;
    case 138: SDF_PAIR_MIXED(8); break;
    case 139: SDF_PAIR_MIXED(9); break;
    case 123: SDF_PAIR_TRACK(3); break;
    case 126: SDF_PAIR_TRACK(6); break;
    case 608: /* chained strips: edge columns and row-0 sums reset first */
      hipLaunchKernelGGL(strip_chain_init_kernel, one, dim3(64), 0, sdp, lp, lo, dir_reg);
      hipLaunchKernelGGL(extz2_strip_chain_kernel<8>, one, dim3(64), 0, sdp, lp, lo, d_pool, sk, dir_reg, d_out, gave_up, spin_cap, claim);
      break;
    case 604:
      hipLaunchKernelGGL(strip_chain_init_kernel, one, dim3(64), 0, sdp, lp, lo, dir_reg);
      hipLaunchKernelGGL(extz2_strip_chain_kernel<4>, one, dim3(64), 0, sdp, lp, lo, d_pool, sk, dir_reg, d_out, gave_up, spin_cap, claim);
      break;
    case 500:
      hipLaunchKernelGGL(extz2_strip_kernel, half, dim3(64), L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out);
      break;
    case 301: SDF_STRIPE(1) break;
    case 302: SDF_STRIPE(2) break;
    case 304: SDF_STRIPE(4) break;
    case 401: SDF_BSTRIPE(1) break;
    case 402: SDF_BSTRIPE(2) break;
    case 404: SDF_BSTRIPE(4) break;
    case 64: SDF_GENERAL(64, false); break;
    case 256: SDF_GENERAL(256, false); break;
    case 1024: SDF_GENERAL(1024, false); break;
    case 2256: SDF_GENERAL(256, true); break;
    case 3024: SDF_GENERAL(1024, true); break;
    case 2001: SDF_GENERAL_HBM(1024, true); break;
    case 1001: SDF_GENERAL_HBM(1024, false); break;
    default: SDF_GENERAL_HBM(256, false); break;  // 1000
  }
#undef SDF_WAVE
#undef SDF_PAIR
#undef SDF_PAIR_MIXED
#undef SDF_PAIR_TRACK
#undef SDF_STRIPE
#undef SDF_BSTRIPE
#undef SDF_GENERAL
#undef SDF_GENERAL_HBM
}

template <int LAYOUT>
static void launch_traceback(bool solo, size_t cnt, hipStream_t s, const PlanTask *lp, const uint32_t *d_pool,
                             const uint8_t *dir_reg, sdf_result *d_out, uint32_t *d_stage) {
  // few tasks: a wavefront per walk (runs of up to 64 cells per round), else four walks per wavefront (16 cells;
  // 8 and 32 cells per walk were measured within noise of 16 on the headline batch)
  if (solo)
    hipLaunchKernelGGL((traceback_kernel<LAYOUT, 64>), dim3((unsigned)cnt), dim3(64), 0, s, lp, (int)cnt, d_pool, dir_reg,
                       d_out, d_stage);
  else
    hipLaunchKernelGGL((traceback_kernel<LAYOUT, 16>), dim3((unsigned)((cnt + 3) / 4)), dim3(64), 0, s, lp, (int)cnt,
                       d_pool, dir_reg, d_out, d_stage);
}

// Uploads chunk `ci` of the cut and launches its DP kernels and its traceback.
// Streams: Q[0] the caller's, Q[1], Q[2] the two DP streams, Q[3] traceback.  The big launches (>= 2048 tasks) of
// ordinary chunks alternate between the DP streams; every other launch (a class of a few tasks ends in a tail as
// long as its longest task) goes, longest first, to the stream with the least estimated work queued.  A heavy chunk
// uploads and traces back on the caller's stream and uses the workspace slice behind the regions.
static int launch_chunk(BatchRun &run, size_t ci) {
  sdf_ctx *ctx = run.ctx;
  const BatchCut &cut = *run.cut;
  const ChunkPlan &c = cut.chunks[ci];
  const size_t cnt = c.cnt, pb = c.pb, ob = c.ob;
  if (cnt == 0) {
    run.cev[ci] = ChunkEv{};
    return SDF_OK;
  }
  hipStream_t st = run.st;
  const bool pipelined = cut.pipelined, heavy_chunk = c.heavy;
  const bool piped = pipelined && !heavy_chunk;
  const size_t nj = run.normal_ids.size();  // ordinal among the ordinary chunks
  const size_t nchunks = cut.chunks.size() + (run.more_chunks ? 1 : 0);
  if (pipelined) {
    // extra streams, created the first time they are wanted (a stream is a hardware queue: ~7 ms to set up): launches of
    // few tasks last as long as their longest task whatever else runs, so the more of them run side by side the better
    // (batches of a few ten thousand tasks, the stage driver's rounds, get by with two: 7 ms per stream is their budget)
    // (a chunk of many launch classes -- a batch of banded tasks of all lengths -- wants three queues of its own next to
    // the three of the chunk before it: see the assignment below)
    const size_t want_aux = nchunks == 1 ? (c.launches.size() > 4 ? std::min<size_t>(c.launches.size() - 4, 4) : 0)
                                         : (cut.ntask_total >= 200000 || c.launches.size() > 4 ? 4 : 2);
    for (size_t a = 0; a < want_aux && a < ctx->aux_limit; ++a)
      if (!ctx->aux_stream[a] && hipStreamCreateWithFlags(&ctx->aux_stream[a], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->aux_stream[a] = nullptr;
      }
  }
  // A chunk of several mixed-pair launches (banded tasks of all lengths, extz2_pair.hip MIXED): each of them ends with its
  // longest chain of rows whatever else runs, so they must START together -- eight more streams, four per chunk parity
  size_t n_mixed_launches = 0;
  for (const Launch &L : c.launches) n_mixed_launches += L.bs >= 130 && L.bs < 140;
  if (pipelined && n_mixed_launches >= 2 && !run.have_heavy)
    for (size_t a = 0; a < 8; ++a)
      if (!ctx->wide_stream[a] && hipStreamCreateWithFlags(&ctx->wide_stream[a], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->wide_stream[a] = nullptr;
      }
  constexpr int NQ = 16;
  hipStream_t Q[NQ] = {st, pipelined ? ctx->dp_stream[0] : st, pipelined ? ctx->dp_stream[1] : st,
                       pipelined ? ctx->tb_stream : st, ctx->aux_stream[0], ctx->aux_stream[1], ctx->aux_stream[2],
                       ctx->aux_stream[3]};
  for (int a = 0; a < 8; ++a) Q[8 + a] = pipelined && !run.have_heavy ? ctx->wide_stream[a] : nullptr;
  // Every internal stream waits for the start of the call before its first use in it -- whatever it is used for (plan
  // upload, DP, traceback): the packed pool's upload, the reset of the result records, the give-up word and whatever the
  // caller ordered on its stream all lie before ev_begin.  (The extra streams are created lazily, above: a fixed list
  // at the start of the call would miss them.)
  for (int q = 1; q < NQ; ++q) {
    if (!Q[q] || Q[q] == st) continue;
    bool seen = false;
    for (size_t b = 0; b < run.nbegan; ++b) seen = seen || run.began[b] == Q[q];
    if (seen) continue;
    SDF_HIP(hipStreamWaitEvent(Q[q], run.ev_begin, 0));
    if (run.nbegan < 24) run.began[run.nbegan++] = Q[q];
  }
  // Q[4..7]: only the least-loaded-stream assignment below uses them (one-chunk batches without heavy tasks)
  // upload stream (and the big launches'): consecutive ordinary chunks alternate between two, so that a chunk's upload
  // and DP do not queue behind the previous chunk's (with heavy tasks in the batch Q[0], Q[1], Q[4], Q[5] are theirs)
  const int ui = piped ? (run.have_heavy ? ((nj & 1) && Q[7] ? 7 : 2) : 1 + (int)(nj & 1)) : 0;
  // (the tracebacks of consecutive ordinary chunks alternate between two streams: the last one starts when its DP
  // ends, not when the previous chunk's walk does)
  // (without the extra streams the call's LAST chunk walks on its own upload / DP stream, which no later chunk waits on:
  // the last two chunks end together when they shared the device, and their walks then run side by side)
  const bool last_chunk = ci + 1 == cut.chunks.size() && !run.more_chunks;
  hipStream_t stb = piped ? ((nj & 1) && ctx->aux_stream[2] ? ctx->aux_stream[2]
                                                             : (last_chunk && nj > 0 && !run.have_heavy ? Q[ui] : Q[3]))
                          : Q[0];
  uint8_t *dir_reg = heavy_chunk ? run.heavy_dir : run.d_dir + cut.heavy_need + (nj % cut.nreg_ws) * cut.region_need;
  hipEvent_t region_ev = nullptr;  // the region's previous user has been traced back
  if (piped && nj >= cut.nreg_ws) region_ev = run.cev[run.normal_ids[nj - cut.nreg_ws]].tb1;
  if (!heavy_chunk) run.normal_ids.push_back(ci);
  SDF_HIP(hipMemcpyAsync(run.d_plan + pb, run.plan + pb, cnt * sizeof(PlanTask), hipMemcpyHostToDevice, Q[ui]));
  SDF_HIP(hipMemcpyAsync(run.d_order + ob, run.order + ob, c.nord * sizeof(int32_t), hipMemcpyHostToDevice, Q[ui]));
  ChunkEv &ev = run.cev[ci];
  ev.dp0 = next_event(ctx, run.evc);
  for (auto &e : ev.dpe) e = nullptr;
  ev.tb0 = next_event(ctx, run.evc);
  ev.tb1 = next_event(ctx, run.evc);
  SDF_HIP(hipEventRecord(ev.dp0, Q[ui]));
  bool used[NQ] = {};
  size_t gs_off = 0;
  {  // HBM state slabs of the very long tasks of this chunk: one allocation, a slice per launch
    size_t gs_total = 0;
    for (const Launch &L : c.launches)
      if (L.bs == 1000 || L.bs == 1001 || L.bs == 2001) gs_total += L.lds * L.cnt;
    if (gs_total > ctx->gstate_buf.cap) {  // growing frees the old slabs: nothing may be using them
      for (hipStream_t q : Q)
        if (q) SDF_HIP(hipStreamSynchronize(q));
      if (ctx->gstate_buf.reserve(gs_total) != hipSuccess) {
        ctx->err = "cannot allocate the HBM state slabs for very long tasks";
        (void)hipGetLastError();
        return SDF_ERR_NOMEM;
      }
    }
  }
  const bool dbg_cls = ctx->cfg.debug_classes != 0;
  if (dbg_cls) {  // what each launch class of the chunk holds: tasks, anti-diagonals, in-band cells (profiles/r04_mm8_classes.txt)
    for (const Launch &L : c.launches) {
      const bool striped = (L.bs >= 300 && L.bs < 500) || L.bs == 604 || L.bs == 608;
      long long nt = 0, rows = 0, cells = 0;
      int32_t prev = -1;
      auto add = [&](int32_t rel) {
        const PlanTask &p = run.plan[pb + rel];
        ++nt;
        rows += p.qlen + p.tlen - 1;
        cells += (p.w >= p.qlen && p.w >= p.tlen) ? (long long)p.qlen * p.tlen : sdf_band_cells(p.qlen, p.tlen, p.w);
      };
      for (size_t j = 0; j < L.cnt; ++j) {
        const int32_t e = run.order[ob + L.off + j];
        if (striped) {
          if (((uint32_t)e >> 24) != 0) continue;  // (one entry per stripe: the first stands for the task)
          const int32_t rel = e & 0xffffff;
          add(rel);
          if (L.bs >= 600 && run.plan[pb + rel].zdrop != rel) add(run.plan[pb + rel].zdrop);  // (chained strips: the partner)
        } else if (e != prev) {  // (a task paired with itself is listed twice)
          add(e);
          prev = e;
        }
      }
      fprintf(stderr, "[class chunk %zu%s bs %d entries %zu tasks %lld rows %lld cells %lld lds %zu]\n", ci, c.heavy ? " (heavy)" : "",
              L.bs, L.cnt, nt, rows, cells, L.lds);
    }
  }
  for (const Launch &L : c.launches) {
    // with heavy tasks in the batch the streams are divided: Q[0], Q[1] for the heavy launches (tens of
    // milliseconds each), the others for the ordinary chunks, which would otherwise queue behind them
    int qi = 0;
    if (pipelined) {
      const bool mixed_cls = L.bs >= 130 && L.bs < 140;  // (long and short banded tasks in one launch: it ends with its longest chain,
                                                         // and runs next to the chunk's other launches rather than behind them)
      if (piped && L.cnt >= 2048 && (L.bs < 300 || L.bs == 500) && !mixed_cls) {  // (a stripe class counts stripes, and lasts as long as its longest task)
        qi = ui;
      } else {  // least estimated work queued; with heavy tasks in the batch Q[0], Q[1], Q[4], Q[5] are theirs
        // Without heavy tasks, consecutive ordinary chunks keep to disjoint sets of queues -- Q[1], Q[4], Q[5] and Q[2],
        // Q[6], Q[7]: a chunk's launches (and its plan upload, which every one of them waits for) must not queue behind
        // the long-running small launches of the chunk before it, or the chunks run one after the other, each as long as
        // its longest task (a 100,000-task batch of banded tasks of all lengths: 474 ms -> see profiles/r03_shapes.txt)
        const bool split_q = !run.have_heavy && piped && nchunks > 1;
        auto mine = [&](int q) { return (nj & 1) ? (q == 2 || q == 6 || q == 7 || q >= 12) : (q == 1 || q == 4 || q == 5 || (q >= 8 && q < 12)); };
        qi = run.have_heavy ? (heavy_chunk ? 0 : 2) : split_q ? ui : 1;
        for (int q = qi + 1; q < NQ; ++q) {
          if (!Q[q] || (run.have_heavy && heavy_chunk != (q == 1 || q == 4 || q == 5))) continue;
          if (split_q && !mine(q)) continue;
          if (run.qload[q] < run.qload[qi]) qi = q;
        }
        if (split_q)  // (the loop above starts at the upload queue: the lower queues of the set)
          for (int q = 1; q < qi; ++q)
            if (Q[q] && mine(q) && run.qload[q] < run.qload[qi]) qi = q;
      }
    }
    run.qload[qi] += L.est;
    hipStream_t sdp = Q[qi];
    if (!used[qi]) {
      used[qi] = true;
      if (pipelined && qi != ui) SDF_HIP(hipStreamWaitEvent(sdp, ev.dp0, 0));  // plan uploaded
      if (region_ev) SDF_HIP(hipStreamWaitEvent(sdp, region_ev, 0));
    }
    uint8_t *slabs = nullptr;
    if (L.bs == 1000 || L.bs == 1001 || L.bs == 2001) {
      slabs = (uint8_t *)ctx->gstate_buf.p + gs_off;
      gs_off += L.lds * L.cnt;
    }
    unsigned *claim = nullptr;  // a stripe launch's entry counters (stripe_claim): one set of eight per launch of the call
    if (((L.bs >= 300 && L.bs < 500) || L.bs == 604 || L.bs == 608) && ctx->stripe_claim && L.cnt % 8 == 0 &&
        run.claim_sets < kClaimSets)
      claim = (unsigned *)ctx->claim_buf.p + 8 * run.claim_sets++;
    launch_dp(L, sdp, run.d_plan + pb, run.d_order + ob + L.off, run.d_pool, run.sk, dir_reg, run.d_out, slabs,
              (unsigned long long *)ctx->misc_buf.p + 1, ctx->stripe_spin_cap, claim);
    if ((L.bs >= 300 && L.bs < 500) || L.bs == 604 || L.bs == 608) run.any_stripe = true;
    ++ctx->launches;
  }
  for (int q = 0; q < NQ; ++q) {  // the traceback stream collects every stream the chunk's DP ran on
    if (!used[q]) continue;
    ev.dpe[q] = next_event(ctx, run.evc);
    SDF_HIP(hipEventRecord(ev.dpe[q], Q[q]));
    if (pipelined && Q[q] != stb) SDF_HIP(hipStreamWaitEvent(stb, ev.dpe[q], 0));
  }
  SDF_HIP(hipEventRecord(ev.tb0, stb));
  if (run.want_cigar) {
    const unsigned layouts = c.layouts;  // one traceback instantiation per direction-flag layout present
    const bool tb_solo = cnt <= (nchunks == 1 ? (size_t)8192 : (size_t)1024);
    // A walk is a few rounds of memory latency per CIGAR run: a launch lasts as long as its longest task.  When a
    // chunk of few tasks mixes layouts, the instantiations run side by side on different streams (each after the
    // chunk's DP, collected again by the chunk's traceback stream) rather than one after the other.
    const bool side_by_side = pipelined && cnt < 32768 && (layouts & (layouts - 1)) != 0;
    int used_tb = 0;
    hipStream_t tbs[3] = {stb, stb, stb};  // (a fourth layout shares the last stream)
    if (side_by_side) {  // (a batch with heavy tasks keeps its stream division: Q[0], Q[1] heavy, Q[2], Q[3] ordinary)
      int j = 1;
      for (int q = 0; q < 4 && j < 3; ++q) {
        if (Q[q] == stb) continue;
        if (run.have_heavy && (heavy_chunk ? q >= 2 : q < 2)) continue;
        tbs[j++] = Q[q];
      }
    }
    auto tb_on = [&]() -> hipStream_t {
      hipStream_t s2 = tbs[used_tb < 3 ? used_tb : 2];
      if (side_by_side && s2 != stb) (void)hipStreamWaitEvent(s2, ev.tb0, 0);
      ++used_tb;
      return s2;
    };
    const PlanTask *lp = run.d_plan + pb;
    if (layouts & 64u) launch_traceback<6>(tb_solo, cnt, tb_on(), lp, run.d_pool, dir_reg, run.d_out, run.d_stage);
    if (layouts & 16u) launch_traceback<4>(tb_solo, cnt, tb_on(), lp, run.d_pool, dir_reg, run.d_out, run.d_stage);
    if (layouts & 8u) launch_traceback<3>(tb_solo, cnt, tb_on(), lp, run.d_pool, dir_reg, run.d_out, run.d_stage);
    if (layouts & 4u) launch_traceback<2>(tb_solo, cnt, tb_on(), lp, run.d_pool, dir_reg, run.d_out, run.d_stage);
    if (layouts & 2u) launch_traceback<1>(tb_solo, cnt, tb_on(), lp, run.d_pool, dir_reg, run.d_out, run.d_stage);
    if (layouts & 1u) launch_traceback<0>(tb_solo, cnt, tb_on(), lp, run.d_pool, dir_reg, run.d_out, run.d_stage);
    if (side_by_side)
      for (int j = 1; j < used_tb && j < 3; ++j) {
        if (tbs[j] == stb) continue;
        hipEvent_t e = next_event(ctx, run.evc);
        SDF_HIP(hipEventRecord(e, tbs[j]));
        SDF_HIP(hipStreamWaitEvent(stb, e, 0));
      }
  }
  SDF_HIP(hipEventRecord(ev.tb1, stb));
  return SDF_OK;
}

// One workgroup per re-run task: its record over the one the stripe kernel left (n_cigar = -1), its CIGAR runs to the end
// of the task's staging slot, where the compaction looks for them (in walk order: a reversed CIGAR is turned back, the
// compaction reverses it again).
struct RerunMap {
  int32_t out_idx, cig_cap, flag, pad;
  int64_t cig_slot;
};
__global__ __launch_bounds__(64) void rerun_merge_kernel(const RerunMap *__restrict__ map, const sdf_result *__restrict__ rr_out,
                                                         const uint32_t *__restrict__ rr_cig, sdf_result *__restrict__ res,
                                                         uint32_t *__restrict__ stage) {
  const RerunMap m = map[blockIdx.x];
  const sdf_result r = rr_out[blockIdx.x];
  const int nc = r.n_cigar;
  if (nc > 0 && nc <= m.cig_cap) {
    const uint32_t *src = rr_cig + r.cigar_off;
    uint32_t *dst = stage + m.cig_slot + (m.cig_cap - nc);
    const bool rev = (m.flag & SDF_FLAG_REV_CIGAR) != 0;
    for (int c = threadIdx.x; c < nc; c += 64) dst[c] = src[rev ? nc - 1 - c : c];
  }
  if (threadIdx.x == 0) {
    sdf_result o = r;
    o.cigar_off = 0;
    res[m.out_idx] = o;
  }
}

// Tasks a stripe kernel gave up (extz2_stripe.hip: stripe_abandon) run again, in this call, on the kernels that keep a
// task inside one wavefront or workgroup: a second context of this device without the stripe kernels aligns them into
// buffers of their own, and their records and CIGAR runs are merged into this batch before its CIGARs are scanned.
static int rerun_abandoned(BatchRun &run, unsigned long long count) {
  sdf_ctx *ctx = run.ctx;
  hipStream_t st = run.st;
  unsigned long long *d_gave = (unsigned long long *)ctx->misc_buf.p + 1;
  if (count > SDF_GAVEUP_CAP) {
    ctx->err = "internal: stripe wavefronts gave up waiting for their neighbours on more tasks than can be re-run";
    return SDF_ERR_INVALID;
  }
  std::vector<uint32_t> idx(count);
  SDF_HIP(hipMemcpyAsync(idx.data(), d_gave + SDF_GAVEUP_LIST, count * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
  std::sort(idx.begin(), idx.end());
  // (a task can stand in the list twice: a block that still held a task's last column may have written the task's
  // record over the mark of a block that had given up, and a third block then marked it again -- ADVICE r3)
  idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
  count = idx.size();
  // their plan entries (staging slots): one pass over the batch's plan
  std::vector<RerunMap> map(count);
  std::vector<sdf_task> sub(count);
  size_t found = 0, cig_cap = 4;
  for (size_t k = 0; k < run.cut->ntask_total; ++k) {
    const PlanTask &p = run.plan[k];
    const auto it = std::lower_bound(idx.begin(), idx.end(), (uint32_t)p.out_idx);
    if (it == idx.end() || *it != (uint32_t)p.out_idx) continue;
    const size_t j = (size_t)(it - idx.begin());
    map[j] = {p.out_idx, p.cig_cap, p.flag, 0, p.cig_slot};
    sub[j] = run.tasks[p.out_idx];
    cig_cap += (size_t)p.cig_cap;
    ++found;
  }
  if (found != count) {
    ctx->err = "internal: give-up list names a task that is not in the plan";
    return SDF_ERR_INVALID;
  }
  if (!ctx->rerun_ctx) {
    // (with the parent's settings, not the environment's -- ADVICE r3 --, minus every kernel that waits for a neighbour)
    sdf_config rc_cfg = ctx->cfg;
    rc_cfg.strip_always = 0;
    rc_cfg.no_stripe = 1;
    rc_cfg.bstripe_min_rows = 0;
    rc_cfg.workspace_gib = 0;
    rc_cfg.debug_plan = 0;
    ctx->rerun_ctx = sdf_create_cfg(ctx->device, ctx->ws_budget / 4, &rc_cfg);
    if (!ctx->rerun_ctx) {
      ctx->err = "cannot create the context that re-runs the tasks a stripe kernel gave up";
      return SDF_ERR_NOMEM;
    }
    // (like a part context: not another user of the process's CPUs -- the split rule and the planner's thread count look at
    // the number of live contexts)
    mark_internal_context(ctx->rerun_ctx);
  }
  SDF_HIP(ctx->rr_out.reserve(count * sizeof(sdf_result)));
  SDF_HIP(ctx->rr_cig.reserve(cig_cap * 4));
  SDF_HIP(ctx->rr_map.reserve(count * sizeof(RerunMap)));
  size_t used = 0;
  const int rc = sdf_extz2_batch_device(ctx->rerun_ctx, run.scoring, sub.data(), count, run.d_pool, run.want,
                                        (sdf_result *)ctx->rr_out.p, (uint32_t *)ctx->rr_cig.p, cig_cap, &used, st);
  if (rc != SDF_OK) {
    ctx->err = std::string("re-run of abandoned stripe tasks: ") + sdf_last_error(ctx->rerun_ctx);
    return rc;
  }
  SDF_HIP(hipMemcpyAsync(ctx->rr_map.p, map.data(), count * sizeof(RerunMap), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rerun_merge_kernel, dim3((unsigned)count), dim3(64), 0, st, (const RerunMap *)ctx->rr_map.p,
                     (const sdf_result *)ctx->rr_out.p, (const uint32_t *)ctx->rr_cig.p, run.d_out, run.d_stage);
  SDF_HIP(hipMemsetAsync(d_gave, 0, sizeof(unsigned long long), st));
  SDF_HIP(hipStreamSynchronize(st));  // (`map` is pageable host memory)
  ctx->reran = (long long)count;
  return SDF_OK;
}

// The lane path is a chain of small kernels in front of its DP launches, next to chain and strip kernels that hold every
// wavefront slot for milliseconds: on the highest queue priority its workgroups are the first to get a slot a finished
// wavefront frees (the hg19-shaped mixture: 12.8 -> 12.2 ms per call under the profiler; SDF_LANE_PRIO=0: a stream like the others).
hipError_t create_lane_stream(const sdf_ctx *ctx, hipStream_t *out) {
  const bool lane_prio = ctx->cfg.lane_prio != 0;
  int prio_lo = 0, prio_hi = 0;
  if (lane_prio && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_hi = 0;
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, lane_prio ? prio_hi : 0);
}

// The lane tasks of a batch (extz2_lane.hip), start to finish on a stream of their own next to the chunks: the records the
// scan wrote -> sort by (class, qlen, tlen) -> CIGAR-slot and flag-region offsets -> plan records behind the host-planned
// ones -> one DP launch per class present -> traceback.  Nothing here waits for the host.
static int launch_lane(BatchRun &run, size_t n) {
  sdf_ctx *ctx = run.ctx;
  const BatchCut &cut = *run.cut;
  const size_t nl = cut.n_lane;
  if (!ctx->lane_stream && create_lane_stream(ctx, &ctx->lane_stream) != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "cannot create the lane kernel's stream";
    return SDF_ERR_HIP;
  }
  hipStream_t sl = ctx->lane_stream;
  const bool dbg_lane = ctx->cfg.debug_plan != 0;
  const auto lt0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (dbg_lane)
      fprintf(stderr, "[lane %s at %.2f ms]\n", what,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lt0).count());
  };
  SDF_HIP(hipStreamWaitEvent(sl, run.ev_begin, 0));
  // Planning without a sort (extz2_lane.hip, "second form"): histogram over the 19-bit keys, one scan over the bins, every task
  // to its rank in its bin.  SDF_LANE_PLAN=sort: round 4's hipCUB radix sort + two scans.
  const bool by_bins = ctx->cfg.lane_plan_sort == 0;
  if (by_bins) {
    constexpr size_t nb = kLaneBins, nt = kLaneBins / kLaneScanBlock;
    SDF_HIP(ctx->ln_recs.reserve(n * sizeof(LaneRec)));
    SDF_HIP(ctx->ln_bins.reserve(nb * (4 + 4 + 4 + 8 + 8) + nt * (8 + 8 + 8) + 256));
    LaneRec *d_recs = (LaneRec *)ctx->ln_recs.p;
    unsigned long long *capbase = (unsigned long long *)ctx->ln_bins.p, *dirbase = capbase + nb, *tot_cap = dirbase + nb, *tot_dir = tot_cap + nt;
    uint32_t *count = (uint32_t *)(tot_dir + nt), *cursor = count + nb, *base = cursor + nb, *tot_cnt = base + nb;
    lap("buffers");
    SDF_HIP(hipMemcpyAsync(d_recs, ctx->host_lane.p, n * sizeof(LaneRec), hipMemcpyHostToDevice, sl));
    SDF_HIP(hipMemsetAsync(count, 0, 2 * nb * sizeof(uint32_t), sl));  // (count and cursor)
    lap("records uploaded");
    const dim3 gn((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(lane_hist_kernel, gn, dim3(256), 0, sl, d_recs, (int)n, count);
    hipLaunchKernelGGL(lane_bins_scan_kernel, dim3((unsigned)nt), dim3(256), 0, sl, count, base, capbase, dirbase, tot_cnt, tot_cap, tot_dir);
    hipLaunchKernelGGL(lane_bins_top_kernel, dim3(1), dim3((unsigned)nt), 0, sl, tot_cnt, tot_cap, tot_dir);
    PlanTask *lp = run.d_plan + cut.ntask_total;
    const int64_t dir0 = (int64_t)(cut.heavy_need + cut.nreg_ws * cut.region_need);
    hipLaunchKernelGGL(lane_place_kernel, gn, dim3(256), 0, sl, d_recs, (int)n, cursor, base, capbase, dirbase, tot_cnt, tot_cap, tot_dir,
                       cut.stage_total, dir0, lp);
    lap("histogram, scan, plan");
    run.ev_lane0 = next_event(ctx, run.evc);
    SDF_HIP(hipEventRecord(run.ev_lane0, sl));
    size_t pos = 0;
    for (int c = 0; c < 4; ++c) {  // (the class is the key's top: the classes are consecutive ranges)
      const size_t cnt = cut.lane_cls[c];
      if (!cnt) continue;
      hipLaunchKernelGGL(extz2_lane_kernel, dim3((unsigned)((cnt + 63) / 64)), dim3(64), lane_lds_bytes(c), sl, lp + pos, (int)cnt,
                         run.d_pool, run.sk, run.d_dir, run.d_out);
      ++ctx->launches;
      pos += cnt;
    }
    if (run.want_cigar)
      launch_traceback<5>(false, nl, sl, lp, run.d_pool, run.d_dir, run.d_out, run.d_stage);
    run.ev_lane = next_event(ctx, run.evc);
    SDF_HIP(hipEventRecord(run.ev_lane, sl));
    SDF_HIP(hipGetLastError());
    ctx->lane_tasks = (long long)nl;
    lap("DP, traceback");
    return SDF_OK;
  }
  SDF_HIP(ctx->ln_recs.reserve(n * sizeof(LaneRec)));
  SDF_HIP(ctx->ln_keys.reserve(n * 8));
  SDF_HIP(ctx->ln_vals.reserve(n * 8));
  SDF_HIP(ctx->ln_sizes.reserve(nl * 32 + 64));
  LaneRec *d_recs = (LaneRec *)ctx->ln_recs.p;
  uint32_t *k_in = (uint32_t *)ctx->ln_keys.p, *k_out = k_in + n, *v_in = (uint32_t *)ctx->ln_vals.p, *v_out = v_in + n;
  unsigned long long *cap = (unsigned long long *)ctx->ln_sizes.p, *dirb = cap + nl, *cap_off = dirb + nl, *dir_off = cap_off + nl;
  size_t t_sort = 0, t_scan = 0;
  SDF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, t_sort, k_in, k_out, v_in, v_out, (int)n, 0, 20, sl));
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t_scan, cap, cap_off, (int)nl, sl));
  const size_t t_bytes = std::max(t_sort, t_scan) + 256;
  SDF_HIP(ctx->ln_tmp.reserve(t_bytes));
  lap("buffers");
  SDF_HIP(hipMemcpyAsync(d_recs, ctx->host_lane.p, n * sizeof(LaneRec), hipMemcpyHostToDevice, sl));
  lap("records uploaded");
  hipLaunchKernelGGL(lane_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sl, d_recs, (int)n, k_in, v_in);
  lap("keys");
  size_t tb = t_bytes;
  SDF_HIP(hipcub::DeviceRadixSort::SortPairs(ctx->ln_tmp.p, tb, k_in, k_out, v_in, v_out, (int)n, 0, 20, sl));
  lap("sort");
  const dim3 gl((unsigned)((nl + 255) / 256));
  hipLaunchKernelGGL(lane_sizes_kernel, gl, dim3(256), 0, sl, d_recs, v_out, (int)nl, cap, dirb);
  tb = t_bytes;
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->ln_tmp.p, tb, cap, cap_off, (int)nl, sl));
  tb = t_bytes;
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->ln_tmp.p, tb, dirb, dir_off, (int)nl, sl));
  // plan records behind the host-planned ones; staging slots behind theirs; flags in the slice behind the heavy tasks'
  PlanTask *lp = run.d_plan + cut.ntask_total;
  const int64_t dir0 = (int64_t)(cut.heavy_need + cut.nreg_ws * cut.region_need);
  hipLaunchKernelGGL(lane_plan_kernel, gl, dim3(256), 0, sl, d_recs, v_out, (int)nl, cap_off, dir_off, cut.stage_total, dir0, lp);
  lap("scans, plan");
  run.ev_lane0 = next_event(ctx, run.evc);
  SDF_HIP(hipEventRecord(run.ev_lane0, sl));
  size_t pos = 0;
  for (int c = 0; c < 4; ++c) {  // (sorted by class first: the classes are consecutive ranges)
    const size_t cnt = cut.lane_cls[c];
    if (!cnt) continue;
    hipLaunchKernelGGL(extz2_lane_kernel, dim3((unsigned)((cnt + 63) / 64)), dim3(64), lane_lds_bytes(c), sl, lp + pos, (int)cnt,
                       run.d_pool, run.sk, run.d_dir, run.d_out);
    ++ctx->launches;
    pos += cnt;
  }
  if (run.want_cigar)
    launch_traceback<5>(false, nl, sl, lp, run.d_pool, run.d_dir, run.d_out, run.d_stage);
  run.ev_lane = next_event(ctx, run.evc);
  SDF_HIP(hipEventRecord(run.ev_lane, sl));
  SDF_HIP(hipGetLastError());
  ctx->lane_tasks = (long long)nl;
  lap("DP, traceback");
  return SDF_OK;
}

// Closes the batch on the caller's stream: waits for every chunk's traceback, scans n_cigar into cigar_off,
// compacts the CIGARs into the caller's pool and reads the timing events.
// every traceback (and the lane tasks) of a part has finished on its stream; the tasks its stripe kernels gave up are run again
static int join_part(BatchRun &run) {
  sdf_ctx *ctx = run.ctx;
  hipStream_t st = run.st;
  if (run.cut->pipelined)
    for (auto &ev : run.cev)
      if (ev.tb1) SDF_HIP(hipStreamWaitEvent(st, ev.tb1, 0));
  if (run.ev_lane) SDF_HIP(hipStreamWaitEvent(st, run.ev_lane, 0));
  if (run.any_stripe) {  // (one more round trip, for batches with stripe launches only)
    unsigned long long gave_up = 0;  // (misc word 1: tasks whose stripe wavefronts gave up waiting for a neighbour)
    SDF_HIP(hipMemcpyAsync(&gave_up, (unsigned long long *)ctx->misc_buf.p + 1, sizeof(gave_up), hipMemcpyDeviceToHost, st));
    SDF_HIP(hipStreamSynchronize(st));
    if (gave_up)
      if (int rc = rerun_abandoned(run, gave_up)) return rc;
  }
  return SDF_OK;
}

// `head`: the part of the call that was launched first, on another context and stream (null: the call is one part);
// n / d_out: the whole call's.
static int finish_batch(BatchRun &run, BatchRun *head, size_t n, sdf_result *d_out, uint32_t *d_cig, size_t cigar_cap,
                        size_t *cigar_used) {
  sdf_ctx *ctx = run.ctx;
  hipStream_t st = run.st;
  if (head) {
    if (int rc = join_part(*head)) {
      ctx->err = head->ctx->err;
      return rc;
    }
    hipEvent_t done = next_event(head->ctx, head->evc);
    SDF_HIP(hipEventRecord(done, head->st));
    SDF_HIP(hipStreamWaitEvent(st, done, 0));
  }
  if (int rc = join_part(run)) return rc;
  unsigned long long *d_total = (unsigned long long *)ctx->misc_buf.p;
  hipEvent_t ev_c0 = next_event(ctx, run.evc), ev_c1 = next_event(ctx, run.evc), ev_end = next_event(ctx, run.evc);
  SDF_HIP(hipEventRecord(ev_c0, st));
  unsigned long long total = 0, gave_up = 0;
  auto plan_records = [](const BatchRun &r) {  // (lane plan records follow the host's)
    return r.cut->ntask_total + (r.cut->use_lane ? r.cut->n_lane : 0);
  };
  if (run.want_cigar) {
    {
      const int nb = (int)((n + 1023) / 1024);
      unsigned long long *d_part = d_total + SDF_MISC_PARTS;
      hipLaunchKernelGGL(cigar_scan_blocks_kernel, dim3((unsigned)nb), dim3(1024), 0, st, d_out, (int)n, d_part);
      hipLaunchKernelGGL(cigar_scan_parts_kernel, dim3(1), dim3(1024), 0, st, d_part, nb, d_total);
      hipLaunchKernelGGL(cigar_scan_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_out, (int)n,
                         (const unsigned long long *)d_part);
    }
    SDF_HIP(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, st));
    SDF_HIP(hipStreamSynchronize(st));
    if (cigar_used) *cigar_used = (size_t)total;
    if (total > cigar_cap || (total && !d_cig)) {
      ctx->err = "CIGAR pool too small";
      return SDF_ERR_CIGAR_OVERFLOW;
    }
    for (const BatchRun *r : {(const BatchRun *)head, (const BatchRun *)&run}) {
      if (!r) continue;
      const size_t np = plan_records(*r);
      if (np)
        hipLaunchKernelGGL(cigar_compact_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, st, r->d_plan, (int)np,
                           r->d_out, r->d_stage, d_cig, (unsigned long long)cigar_cap);
    }
  }
  SDF_HIP(hipEventRecord(ev_c1, st));
  SDF_HIP(hipEventRecord(ev_end, st));
  SDF_HIP(hipMemcpyAsync(&gave_up, d_total + 1, sizeof(gave_up), hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
  SDF_HIP(hipGetLastError());
  if (gave_up) {
    ctx->err = "internal: a stripe wavefront gave up waiting for its neighbour";
    return SDF_ERR_INVALID;
  }
  // DP / traceback time = length of the union of the chunks' intervals (chunks overlap when pipelined);
  // ms[6] = sum of the chunks' DP intervals (what a kernel trace adds up)
  auto span = [&](bool tb, float &sum) {
    std::vector<std::pair<float, float>> iv;
    sum = 0.f;
    auto add = [&](hipEvent_t e0, hipEvent_t e1) {
      float a = 0, b = 0;
      (void)hipEventElapsedTime(&a, run.ev_begin, e0);
      (void)hipEventElapsedTime(&b, run.ev_begin, e1);
      iv.push_back({a, b});
      sum += b - a;
    };
    for (auto &ev : run.cev) {
      if (!ev.dp0) continue;
      if (tb) {
        add(ev.tb0, ev.tb1);
      } else {
        for (hipEvent_t e : ev.dpe)
          if (e) add(ev.dp0, e);
      }
    }
    std::sort(iv.begin(), iv.end());
    float len = 0, end = -1e30f;
    for (auto &p : iv) {
      if (p.first > end) {
        len += p.second - p.first;
        end = p.second;
      } else if (p.second > end) {
        len += p.second - end;
        end = p.second;
      }
    }
    return len;
  };
  const bool dbg_iv = ctx->cfg.debug_plan != 0;
  if (dbg_iv) {  // the chunks' DP intervals per stream and the lane kernel's, from the start of the call
    for (size_t ci = 0; ci < run.cev.size(); ++ci) {
      const ChunkEv &ev = run.cev[ci];
      if (!ev.dp0) continue;
      for (int q = 0; q < 16; ++q)
        if (ev.dpe[q]) {
          float a = 0, b = 0;
          (void)hipEventElapsedTime(&a, run.ev_begin, ev.dp0);
          (void)hipEventElapsedTime(&b, run.ev_begin, ev.dpe[q]);
          fprintf(stderr, "[chunk %zu%s stream %d: DP %.2f - %.2f ms]\n", ci, run.cut->chunks[ci].heavy ? " (heavy)" : "", q, a, b);
        }
    }
    if (run.ev_lane0 && run.ev_lane) {
      float a = 0, b = 0;
      (void)hipEventElapsedTime(&a, run.ev_begin, run.ev_lane0);
      (void)hipEventElapsedTime(&b, run.ev_begin, run.ev_lane);
      fprintf(stderr, "[lane kernel: DP + traceback %.2f - %.2f ms]\n", a, b);
    }
  }
  float s0 = 0, s1 = 0;
  ctx->ms[0] = span(false, s0);
  ctx->ms[1] = span(true, s1);
  ctx->ms[6] = s0;
  (void)hipEventElapsedTime(&ctx->ms[2], ev_c0, ev_c1);
  (void)hipEventElapsedTime(&ctx->ms[3], run.ev_begin, ev_end);
  return SDF_OK;
}

}  // namespace sdf
