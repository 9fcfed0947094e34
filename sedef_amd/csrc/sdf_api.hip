// The C ABI (include/sedef_hip.h): context, packing helpers and the batch entry points.
// Replaces the call site of ksw_extz2_sse in align_helper (reference: src/align.cc:39-68) with a
// batched device path.  No CPU fallback exists here: every DP cell is computed by a gfx950 kernel.
// Planning lives in sdf_plan.hip, uploads and launches in sdf_launch.hip (same translation unit, see sdf_unity.hip).
#include <hip/hip_runtime.h>

#include <sched.h>

#include <atomic>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "sdf_ctx.h"

using namespace sdf;

namespace {
std::string g_err;  // error of the last failed sdf_create
}  // namespace

extern "C" int sdf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *sdf_last_error(const sdf_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

namespace {
std::atomic<int> g_live_contexts{0};
// CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota
int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : CPU_COUNT(&set), CPU_COUNT(&set));
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
    char q[32];
    long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
      n = std::min(n, (int)((atol(q) + period / 2) / period));
    fclose(f);
  }
  return std::max(n, 1);
}
}  // namespace

// the context's fields the planner and the launcher read, from its configuration (one place; the context's settings do
// not change afterwards)
static void apply_config(sdf_ctx *ctx) {
  const sdf_config &c = ctx->cfg;
  ctx->force_general = c.force_general != 0;
  ctx->no_pair = c.no_pair != 0;
  ctx->no_mixed = c.no_mixed != 0;
  ctx->mixed_min = (size_t)c.mixed_min;
  ctx->stripe_claim = c.stripe_claim != 0;
  ctx->chain_min = (size_t)c.chain_min;
  ctx->self_pair_max = (size_t)c.self_pair_max;
  ctx->stats_items = (unsigned)c.stats_items;
  ctx->no_stripe = c.no_stripe != 0;
  ctx->stripe_min = (int)c.stripe_min;
  ctx->bstripe_min_rows = (int)c.bstripe_min_rows;
  ctx->stripe_spin_cap = (int)c.stripe_spin_cap;
  ctx->strip_enabled = c.no_strip == 0;
  ctx->strip_always = c.strip_always != 0;
  ctx->strip_cols = (int)c.strip_cols;
  ctx->lane_enabled = c.no_lane == 0;
  ctx->lane_min = (size_t)c.lane_min;
  ctx->pipeline = c.pipeline != 0;
  if (c.debug_timing) g_debug_timing.store(true, std::memory_order_relaxed);
}

extern "C" const sdf_config *sdf_get_config(const sdf_ctx *ctx) { return ctx ? &ctx->cfg : nullptr; }

extern "C" sdf_ctx *sdf_create(int device, size_t workspace_bytes) { return sdf_create_cfg(device, workspace_bytes, nullptr); }

extern "C" sdf_ctx *sdf_create_cfg(int device, size_t workspace_bytes, const sdf_config *cfg_in) {
  sdf_config cfg;
  if (cfg_in) {
    if (cfg_in->size != sizeof(sdf_config)) {
      g_err = "sdf_create_cfg: the configuration was not initialised by sdf_config_default / sdf_config_from_env (size field)";
      return nullptr;
    }
    cfg = *cfg_in;
  } else {
    char why[256];
    if (sdf_config_from_env(&cfg, why, sizeof why) != SDF_OK) {  // (a typo in an SDF_* variable is an error, not a silent default)
      g_err = std::string("environment: ") + why;
      return nullptr;
    }
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_err = "no HIP device available (this library has no CPU fallback)";
    return nullptr;
  }
  if (device < 0 || device >= n) {
    g_err = "device ordinal out of range";
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_err = "hipSetDevice failed";
    return nullptr;
  }
  const auto t_create = std::chrono::steady_clock::now();
  sdf_ctx *ctx = new sdf_ctx();
  ctx->device = device;
  ctx->cfg = cfg;
  apply_config(ctx);
  if (cfg.debug_plan) {
    std::string dump(sdf_config_dump(&cfg, nullptr, 0), '\0');
    sdf_config_dump(&cfg, &dump[0], dump.size());
    fprintf(stderr, "[sdf_create device %d: configuration]\n%s", device, dump.c_str());
  }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    g_err = "hipStreamCreate failed";
    delete ctx;
    return nullptr;
  }
  auto lap = [&](const char *what) {
    if (cfg.debug_timing)
      fprintf(stderr, "[sdf_create %s at %.1f ms]\n", what,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count());
  };
  lap("first stream");
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  lap("mem info");
  // (0: half of the free HBM -- the workspace is allocated by NEED, region by region (cut_batch), so a large budget costs a
  // small batch nothing, and a batch of long banded tasks -- BASELINE configs[4] at 100,000 tasks: 131 GB of flags -- is not cut
  // into more, smaller chunks because of a constructor default: 64 GiB until round 4, 252 ms against 201 at 128 GiB)
  size_t budget = workspace_bytes ? workspace_bytes : free_b ? free_b / 2 : (size_t)64 << 30;
  if (cfg.workspace_gib > 0) budget = (size_t)(cfg.workspace_gib * 1073741824.0);  // (overrides the caller's figure: experiments with the stage driver)
  if (free_b && budget > free_b / 2) budget = free_b / 2;
  ctx->ws_budget = budget;
  // allow the general kernel its full 160 KiB of LDS
  const int want_lds = 160 * 1024;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_general_kernel<64, false, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, want_lds) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_general_kernel<256, false, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, want_lds) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_general_kernel<1024, false, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, want_lds) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_general_kernel<1024, false, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, want_lds) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_general_kernel<256, false, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, want_lds) == hipSuccess)
    ctx->max_dyn_lds = want_lds;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<1, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<1, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<2, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<2, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<3, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<3, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<6, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<6, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<4, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<4, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<8, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<8, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
#define SDF_PAIR_ATTR(N)                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_pair_kernel<N, false, false>),  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);         \
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_pair_kernel<N, true, false>),   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  SDF_PAIR_ATTR(1) SDF_PAIR_ATTR(2) SDF_PAIR_ATTR(3) SDF_PAIR_ATTR(4) SDF_PAIR_ATTR(6) SDF_PAIR_ATTR(8)
#undef SDF_PAIR_ATTR
  for (const void *f : {reinterpret_cast<const void *>(&extz2_pair_kernel<3, true, true>),
                        reinterpret_cast<const void *>(&extz2_pair_kernel<6, true, true>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<2>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<3>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<4>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<5>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<6>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<8>),
                        reinterpret_cast<const void *>(&extz2_pair_mixed_kernel<9>)})
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipGetLastError();
  for (const void *f : {reinterpret_cast<const void *>(&extz2_stripe_kernel<1>),
                        reinterpret_cast<const void *>(&extz2_stripe_kernel<2>),
                        reinterpret_cast<const void *>(&extz2_stripe_kernel<4>),
                        reinterpret_cast<const void *>(&extz2_bstripe_kernel<1>),
                        reinterpret_cast<const void *>(&extz2_bstripe_kernel<2>),
                        reinterpret_cast<const void *>(&extz2_bstripe_kernel<4>)})
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipGetLastError();
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_lane_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            want_lds);
  // (per context, hence per device: a process-wide once-flag would leave a second GPU's copy of the kernel at 64 KiB)
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sdf::chain_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            std::max(ctx->max_dyn_lds, 65536));
  (void)hipGetLastError();
  lap("attributes");
  if (hipStreamCreateWithFlags(&ctx->dp_stream[0], hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->dp_stream[1], hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->tb_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pipeline = false;
  }
  // (three streams of our own: the runtime multiplexes streams onto GPU_MAX_HW_QUEUES -- default 4 -- hardware
  // queues, and two of ours landing on one queue serialises what the pipeline wants side by side; with the
  // caller's stream that makes four)
  g_live_contexts.fetch_add(1);
  if (cfg.debug_timing)
    fprintf(stderr, "[sdf_create device %d: %.1f ms]\n", device,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count());
  return ctx;
}

void mark_internal_context(sdf_ctx *c) {
  if (!c || c->is_part) return;
  c->is_part = true;
  g_live_contexts.fetch_sub(1);
}

extern "C" void sdf_destroy(sdf_ctx *ctx) {
  if (!ctx) return;
  if (!ctx->is_part) g_live_contexts.fetch_sub(1);
  (void)hipSetDevice(ctx->device);
  for (hipStream_t q : {ctx->stream, ctx->dp_stream[0], ctx->dp_stream[1], ctx->tb_stream, ctx->aux_stream[0],
                        ctx->aux_stream[1], ctx->aux_stream[2], ctx->aux_stream[3]})
    if (q) (void)hipStreamSynchronize(q);
  for (hipStream_t q : ctx->wide_stream)
    if (q) (void)hipStreamSynchronize(q);
  for (auto ev : ctx->events) (void)hipEventDestroy(ev);
  for (DevBuf *b : {&ctx->an_pool, &ctx->an_pairs, &ctx->an_keys, &ctx->an_keys2, &ctx->an_q, &ctx->an_off, &ctx->an_flag,
                    &ctx->an_pos, &ctx->an_cand, &ctx->an_out, &ctx->an_tmp, &ctx->an_outoff, &ctx->ch_an, &ctx->ch_off,
                    &ctx->ch_wsoff, &ctx->ch_work, &ctx->ch_path, &ctx->ch_bounds, &ctx->ch_nb, &ctx->ch_which, &ctx->st_tasks, &ctx->st_pool,
                    &ctx->st_cig, &ctx->st_out})
    b->release();
  for (DevBuf *b : {&ctx->dir_ws, &ctx->stage_ws, &ctx->plan_buf, &ctx->order_buf, &ctx->misc_buf, &ctx->gstate_buf,
                    &ctx->h_pool, &ctx->h_out, &ctx->h_brief, &ctx->h_cig, &ctx->rr_out, &ctx->rr_cig, &ctx->rr_map, &ctx->ln_recs,
                    &ctx->ln_keys, &ctx->ln_vals, &ctx->ln_sizes, &ctx->ln_tmp})
    b->release();
  if (ctx->lane_stream) (void)hipStreamDestroy(ctx->lane_stream);
  ctx->host_lane.release();
  ctx->host_an.release();
  ctx->host_chars.release();
  ctx->pk_recs.release();
  if (ctx->rerun_ctx) sdf_destroy(ctx->rerun_ctx);
  if (ctx->part_ctx) sdf_destroy(ctx->part_ctx);
  if (ctx->part_ev) (void)hipEventDestroy(ctx->part_ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  for (hipStream_t s : {ctx->dp_stream[0], ctx->dp_stream[1], ctx->tb_stream, ctx->aux_stream[0], ctx->aux_stream[1],
                        ctx->aux_stream[2], ctx->aux_stream[3]})
    if (s) (void)hipStreamDestroy(s);
  for (hipStream_t s : ctx->wide_stream)
    if (s) (void)hipStreamDestroy(s);
  if (!ctx->pool_shared) delete ctx->pool;
  delete ctx->cut;
  ctx->host_plan.release();
  ctx->host_order.release();
  ctx->host_pool.release();
  ctx->host_out.release();
  delete ctx;
}

extern "C" size_t sdf_packed_words(int32_t len) {
  if (len <= 0) return 0;
  return (size_t)(len + 15) / 16 + (size_t)(len + 31) / 32;
}

extern "C" void sdf_pack_codes(const uint8_t *codes, int32_t len, uint32_t *out) {
  const size_t nw = sdf_packed_words(len);
  std::memset(out, 0, nw * sizeof(uint32_t));
  uint32_t *cw = out, *nm = out + (len + 15) / 16;
  int32_t k = 0;
  // eight bases at a time: a code >= 4 is N (mask bit set, code bits 0); codes are 0..4 in the callers, any
  // byte >= 4 is treated as N like in the byte loop below
  for (; k + 8 <= len; k += 8) {
    uint64_t x;
    std::memcpy(&x, codes + k, 8);
    // bytes >= 4 have one of bits 2..7 set: fold them onto bit 0 of each byte
    uint64_t hi = x & 0xFCFCFCFCFCFCFCFCull;
    hi |= hi >> 4;
    hi |= hi >> 2;
    hi |= hi >> 1;
    const uint64_t nb = hi & 0x0101010101010101ull;
    const uint32_t nbits = (uint32_t)((nb * 0x0102040810204080ull) >> 56);
    uint64_t y = x & 0x0303030303030303ull & ~(nb * 0xFFull);
    y = (y | (y >> 6)) & 0x000F000F000F000Full;
    y = (y | (y >> 12)) & 0x000000FF000000FFull;
    y = (y | (y >> 24)) & 0xFFFFull;
    cw[k >> 4] |= (uint32_t)y << ((k & 15) * 2);
    nm[k >> 5] |= nbits << (k & 31);
  }
  for (; k < len; ++k) {
    const uint8_t c = codes[k];
    if (c >= 4) nm[k >> 5] |= 1u << (k & 31);
    else cw[k >> 4] |= (uint32_t)c << ((k & 15) * 2);
  }
}

extern "C" size_t sdf_pack_tasks(const uint8_t *codes, const int64_t *q_off, const int32_t *qlen, const int64_t *t_off,
                                 const int32_t *tlen, size_t n, uint32_t *out, int64_t *q_word, int64_t *t_word) {
  size_t words = 0;
  for (size_t k = 0; k < n; ++k) {
    q_word[k] = (int64_t)words;
    words += sdf_packed_words(qlen[k]);
    t_word[k] = (int64_t)words;
    words += sdf_packed_words(tlen[k]);
  }
  if (out)
    for (size_t k = 0; k < n; ++k) {
      if (qlen[k] > 0) sdf_pack_codes(codes + q_off[k], qlen[k], out + q_word[k]);
      if (tlen[k] > 0) sdf_pack_codes(codes + t_off[k], tlen[k], out + t_word[k]);
    }
  return words;
}

extern "C" int64_t sdf_band_cells(int32_t qlen, int32_t tlen, int32_t w) {
  if (qlen <= 0 || tlen <= 0) return 0;
  if (w < 0) w = tlen > qlen ? tlen : qlen;
  int64_t cells = 0;
  for (int r = 0; r < qlen + tlen - 1; ++r) {
    Band b;
    if (!band_of(r, qlen, tlen, w, b)) break;
    cells += b.hi0 - b.lo0 + 1;
  }
  return cells;
}

extern "C" float sdf_last_ms(const sdf_ctx *ctx, int which) {
  if (!ctx || which < 0 || which > 6) return 0.f;
  return ctx->ms[which];
}

extern "C" int sdf_last_launches(const sdf_ctx *ctx) { return ctx ? ctx->launches : 0; }
extern "C" long long sdf_last_paired(const sdf_ctx *ctx) { return ctx ? ctx->paired : 0; }
extern "C" long long sdf_last_reran(const sdf_ctx *ctx) { return ctx ? ctx->reran : 0; }
extern "C" long long sdf_last_lane_tasks(const sdf_ctx *ctx) { return ctx ? ctx->lane_tasks : 0; }

namespace {

int make_scorek(sdf_ctx *ctx, const sdf_scoring *sc, ScoreK &k, bool &degenerate) {
  if (!sc || sc->m != 5) {
    ctx->err = "scoring: the GPU path implements the 5-letter alphabet (ACGT + wildcard) only";
    return SDF_ERR_UNSUPPORTED;
  }
  const int q = sc->gapo, e = sc->gape;
  k.q = q;
  k.e = e;
  k.qe = q + e;
  k.q_b = (uint8_t)q;
  k.qe2_b = (uint8_t)((q + e) * 2);
  k.cap_b = (uint8_t)(int8_t)(sc->mat[0] + (q + e) * 2);
  k.sc_match = (uint8_t)sc->mat[0];
  k.sc_mis = (uint8_t)sc->mat[1];
  k.wild = (uint8_t)(sc->m - 1);
  memcpy(k.mat, sc->mat, 25);
  int min_sc = sc->mat[1];
  for (int t = 1; t < sc->m * sc->m; ++t) min_sc = std::min<int>(min_sc, sc->mat[t]);
  degenerate = -min_sc > 2 * (q + e);  // reference returns before any work (:81)
  return SDF_OK;
}

// The scoring's share of the planning environment (the batch entry points and sdf_debug_plan).
// The register-resident window kernels take three differences of the recurrence with 32-bit subtracts (extz2_wave.hip:
// SDF_CORE), which needs every fresh score byte z0 = score + 2 (q + e) in q .. 127: SEDEF's scoring and every sane one.
// Anything else -- bytes that wrap, a mismatch below -(q + 2 e) -- runs on the general kernel, which emulates the reference's
// bytes one by one.
static void scoring_gates(const sdf_scoring *sc, PlanEnv &env) {
  const int qe2 = 2 * (sc->gapo + sc->gape), zm = sc->mat[0] + qe2, zx = sc->mat[1] + qe2;
  const bool core32_ok = sc->gapo >= 0 && sc->gape >= 0 && zm <= 127 && zx <= 127 && zm >= sc->gapo && zx >= sc->gapo;
  if (!core32_ok) env.force_general = true;
}

// Plans the chunks of a cut, in launch order, on `nthreads` worker threads; wait(ci) blocks until chunk ci is planned.
// With nthreads == 0 wait(ci) plans the chunk itself (small batches: nothing to overlap with).
class ChunkPlanner {
 public:
  // nworkers jobs on `pool` (may be null when nworkers == 0)
  // (`first`: the chunks before it have been planned and launched already -- the early start of the heavy chunks)
  ChunkPlanner(const PlanEnv &env, BatchCut &cut, PlanTask *plan, int32_t *order, WorkerPool *pool, int nworkers, size_t first = 0)
      : env_(env), cut_(cut), plan_(plan), order_(order), pool_(nworkers > 0 ? pool : nullptr),
        ready_(cut.chunks.size(), 0), first_(first) {
    if (!pool_) return;
    next_.store(first + 1);  // chunk `first` is planned by the caller of wait(first): no hand-over in front of the GPU's start
    for (int t = 0; t < nworkers; ++t) pool_->submit([this] { work(); });
  }
  ~ChunkPlanner() {
    stop_.store(true);
    if (pool_) pool_->wait_idle();
  }
  void wait(size_t ci) {
    if (!pool_ || ci == first_) {
      plan_chunk(env_, cut_, cut_.chunks[ci], plan_, order_, own_);
      return;
    }
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] { return ready_[ci] != 0; });
  }

 private:
  void work() {
    PlanScratch sx;
    for (;;) {
      const size_t ci = next_.fetch_add(1);
      if (ci >= cut_.chunks.size() || stop_.load()) return;
      plan_chunk(env_, cut_, cut_.chunks[ci], plan_, order_, sx);
      {
        std::lock_guard<std::mutex> g(mu_);
        ready_[ci] = 1;
      }
      cv_.notify_all();
    }
  }
  const PlanEnv &env_;
  BatchCut &cut_;
  PlanTask *plan_;
  int32_t *order_;
  WorkerPool *pool_;
  std::vector<char> ready_;
  size_t first_ = 0;
  std::atomic<size_t> next_{0};
  std::atomic<bool> stop_{false};
  std::mutex mu_;
  std::condition_variable cv_;
  PlanScratch own_;
};

}  // namespace

// One part of a batch call on one context: cut, plan and launch `n` tasks whose results go to d_out[0 .. n); nothing is
// waited for.  `n_scan`: the records the closing CIGAR scan of this context will cover (the whole call's, when this part
// closes it).  The caller closes the call with finish_batch.
static int batch_part(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, size_t n_scan, const uint32_t *d_pool,
                      uint32_t want, sdf_result *d_out, hipStream_t st, BatchRun &run) {
  ctx->err.clear();
  for (float &m : ctx->ms) m = 0.f;
  ctx->launches = 0;
  ctx->paired = 0;
  // (nothing of this context's earlier calls is in flight: what they outgrew is idle now -- sdf_ctx.h: DevBuf)
  for (DevBuf *b : {&ctx->dir_ws, &ctx->stage_ws, &ctx->plan_buf, &ctx->order_buf, &ctx->gstate_buf, &ctx->h_pool, &ctx->h_out,
                    &ctx->h_brief, &ctx->h_cig, &ctx->ln_recs, &ctx->ln_keys, &ctx->ln_vals, &ctx->ln_sizes, &ctx->ln_tmp})
    b->new_call();
  const auto host_t0 = std::chrono::steady_clock::now();
  auto host_ms = [&] { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count(); };
  run.ctx = ctx;
  run.st = st;
  run.d_pool = d_pool;
  run.d_out = d_out;

  // ---- validate, cut into chunks ----
  PlanEnv env;
  env.cfg = &ctx->cfg;
  env.tasks = tasks;
  env.n = n;
  env.want = want;
  env.want_cigar = (want & SDF_WANT_CIGAR) != 0;
  if (int rc = make_scorek(ctx, sc, run.sk, env.degenerate)) return rc;
  env.gapo = sc->gapo;
  env.max_dyn_lds = ctx->max_dyn_lds;
  env.force_general = ctx->force_general;
  scoring_gates(sc, env);
  env.no_pair = ctx->no_pair;
  env.self_pair_max = ctx->self_pair_max;
  env.no_mixed = ctx->no_mixed;
  env.mixed_min = ctx->mixed_min;
  env.no_stripe = ctx->no_stripe;
  env.stripe_min = ctx->stripe_min;
  env.bstripe_min_rows = ctx->bstripe_min_rows;
  // lane kernel: a tame scoring (every byte of the reference's state stays in 0..127: no wrap-around, no signed /
  // unsigned or sign-extension artefacts) and nothing but CIGAR / score / mte wanted
  {
    const int qe2 = 2 * (sc->gapo + sc->gape), zm = sc->mat[0] + qe2, zx = sc->mat[1] + qe2;
    env.lane_ok = ctx->lane_enabled && ctx->pipeline && !ctx->force_general && !env.degenerate && !(want & SDF_WANT_EXT) &&
                  sc->gapo >= 0 && sc->gape >= 0 && zm >= 0 && zm <= 127 && zx >= 0 && zx <= 127 && n >= ctx->lane_min;
    env.lane_min = ctx->lane_min;
    env.strip_always = ctx->strip_always;
    env.strip_cols = ctx->strip_cols;
    env.chain_min = ctx->chain_min;
    // (zx - q >= 0: the strip kernels take z - q with a 32-bit subtract on packed halves, extz2_strip.hip)
    env.strip_ok = ctx->strip_enabled && !ctx->force_general && !env.degenerate && sc->gapo >= 0 && sc->gape >= 0 && zm >= 0 &&
                   zm <= 127 && zx >= 0 && zx <= 127 && zx - sc->gapo >= 0 && sc->mat[0] >= 0;
    if (env.lane_ok) {
      SDF_HIP(ctx->host_lane.reserve(n * sizeof(LaneRec)));
      env.lane_recs = (LaneRec *)ctx->host_lane.p;
    }
  }
  ctx->lane_tasks = 0;
  run.want_cigar = env.want_cigar;
  run.scoring = sc;
  run.tasks = tasks;
  run.want = want;
  ctx->reran = 0;
  if (!ctx->cut) ctx->cut = new BatchCut();
  BatchCut &cut = *ctx->cut;
  cut.reset();
  {
    // Planning threads of a context: SDF_PLAN_THREADS, else every CPU the process may use but this thread's when the
    // context is the only one of the process (the scan of a million tasks is CPU-bound: 1.7 ms on eight threads, 0.86 ms on
    // sixteen), seven when there are several (the stage driver's lanes share the machine).
    const int env_planners = (int)ctx->cfg.plan_threads;  // (-1: by the CPUs)
    // (one process per GPU: the ranks of a node share its CPUs -- LOCAL_WORLD_SIZE / WORLD_SIZE as torch.distributed sets them)
    static const int local_ranks = [] {
      const char *e = getenv("LOCAL_WORLD_SIZE");
      if (!e) e = getenv("WORLD_SIZE");
      return e ? std::max(1, atoi(e)) : 1;
    }();
    // (several contexts in one process AND several processes on the node -- bench.py's two calls in flight per rank: seven, or
    // the rank's share of the CPUs if that is less)
    const int share = std::max(1, std::min(15, usable_cpus() / local_ranks - 1));
    const int max_planners = env_planners >= 0 ? env_planners : g_live_contexts.load() > 1 ? std::min(7, share) : share;
    // (parked threads plan the chunks of batches of 120,000 tasks and more -- 250,000 tasks of the hg19 mixture:
    // 14.5 -> 10.1 ms, the headline batch unchanged -- and scan the cut as well: sdf_plan.hip, scan_from)
    const size_t pool_from = (size_t)ctx->cfg.plan_pool_from;
    if (!ctx->pool && n >= pool_from && max_planners > 0 && !ctx->is_part) ctx->pool = new WorkerPool(max_planners, (int)ctx->cfg.pool_spin_us);
  }
  run.cut = &cut;
  const bool dbg_plan_chunks = ctx->cfg.debug_plan != 0;

  // ---- buffers and the start of the call on the device: once with upper bounds, before the early start of the heavy
  // chunks (while the batch is still being read), and / or with the cut's sums ----
  bool begun = false;
  auto prepare = [&](const bool early) -> int {
    const size_t lane_need = cut.use_lane ? ((cut.lane_dir_bytes + 255) & ~(size_t)255) : 0;
    // (heavy tasks' slice first: it stays where it is when the workspace has to grow after the early start -- an
    // outgrown buffer is retired, not freed -- and the chunks launched by then keep their pointer)
    const size_t dir_bytes = early ? cut.heavy_need : cut.heavy_need + cut.region_need * cut.nreg_ws + lane_need;
    if (ctx->dir_ws.reserve(std::max<size_t>(dir_bytes, 256), ctx->ws_budget + 4096) != hipSuccess) {
      ctx->err = "cannot allocate the direction-matrix workspace";
      (void)hipGetLastError();
      return SDF_ERR_NOMEM;
    }
    const size_t n_lane = cut.use_lane ? cut.n_lane : 0;
    const size_t np = early ? n : std::max<size_t>(cut.ntask_total, 1);
    const size_t np_dev = early ? n : np + n_lane;
    const int64_t words = early ? cut.stage_upper : cut.stage_total + (cut.use_lane ? cut.lane_stage_words : 0);
    const size_t nord = early ? std::max<size_t>(cut.order_upper, 2) : std::max<size_t>(cut.order_total, 2);
    SDF_HIP(ctx->stage_ws.reserve((size_t)std::max<int64_t>(words, 4) * 4));
    SDF_HIP(ctx->plan_buf.reserve(np_dev * sizeof(PlanTask)));
    SDF_HIP(ctx->order_buf.reserve(nord * sizeof(int32_t)));
    SDF_HIP(ctx->misc_buf.reserve(SDF_MISC_PARTS * 8 + ((std::max(n, n_scan) + 1023) / 1024 + 1) * 8));
    SDF_HIP(ctx->host_plan.reserve(np * sizeof(PlanTask)));
    SDF_HIP(ctx->host_order.reserve(nord * sizeof(int32_t)));
    // (the plan records and the CIGAR staging of the chunks started early are read again at the end of the call -- the
    // traceback's counters, the compaction, a re-run of abandoned tasks: their buffers were sized by bounds that hold.  The
    // launch order is not: a chunk's segment is uploaded and read by its own launches only, and an outgrown buffer --
    // device or pinned -- stays alive until the context goes, so the order buffers may grow here.)
    if (begun && (run.plan != (PlanTask *)ctx->host_plan.p || run.d_plan != (PlanTask *)ctx->plan_buf.p ||
                  run.d_stage != (uint32_t *)ctx->stage_ws.p)) {
      ctx->err = "internal: a buffer sized by its upper bound had to grow after the early start";
      return SDF_ERR_INVALID;
    }
    run.plan = (PlanTask *)ctx->host_plan.p;  // pinned: the uploads are asynchronous
    run.order = (int32_t *)ctx->host_order.p;
    run.d_plan = (PlanTask *)ctx->plan_buf.p;
    run.d_order = (int32_t *)ctx->order_buf.p;
    run.d_dir = (uint8_t *)ctx->dir_ws.p;
    run.d_stage = (uint32_t *)ctx->stage_ws.p;
    if (!begun) {
      run.heavy_dir = run.d_dir;
      SDF_HIP(ctx->claim_buf.reserve(kClaimSets * 8 * sizeof(unsigned)));
      SDF_HIP(hipMemsetAsync(ctx->claim_buf.p, 0, kClaimSets * 8 * sizeof(unsigned), st));
      SDF_HIP(hipMemsetAsync((unsigned long long *)ctx->misc_buf.p + 1, 0, sizeof(unsigned long long), st));
      run.ev_begin = next_event(ctx, run.evc);
      hipLaunchKernelGGL(reset_results_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_out, (int)n);
      SDF_HIP(hipEventRecord(run.ev_begin, st));
      // (the internal streams are ordered behind ev_begin by launch_chunk, each before its first use in this call)
      begun = true;
    }
    return SDF_OK;
  };
  // the early start: plan and launch the heavy chunks (cut.chunks at that moment) on this thread
  PlanScratch early_scratch;
  const std::function<int()> early = [&]() -> int {
    if (int prc = prepare(true)) return prc;
    run.have_heavy = true;
    run.more_chunks = true;
    run.cev.assign(cut.chunks.size(), ChunkEv{});
    for (size_t ci = 0; ci < cut.chunks.size(); ++ci) {
      ChunkPlan &c = cut.chunks[ci];
      plan_chunk(env, cut, c, run.plan, run.order, early_scratch);
      if (ci == 0) ctx->ms[4] = host_ms();
      if (c.err) {
        ctx->err = c.err;
        return SDF_ERR_INVALID;
      }
      ctx->paired += c.paired;
      const float tw = host_ms();
      if (int lrc = launch_chunk(run, ci)) return lrc;
      if (dbg_plan_chunks) fprintf(stderr, "[chunk %zu: %zu tasks (heavy, early) planned by %.2f ms, launched by %.2f ms]\n", ci, c.cnt, tw, host_ms());
      cut.n_early = ci + 1;
    }
    return SDF_OK;
  };
  {
    const char *msg = nullptr;
    const bool early_on = ctx->cfg.early_heavy != 0;  // (0: the heavy chunks wait for the whole cut, as every other chunk)
    const int crc = cut_batch(env, ctx->pipeline, ctx->ws_budget, cut, &msg, ctx->pool, early_on && !ctx->is_part ? &early : nullptr);
    run.more_chunks = false;
    if (crc) {
      if (ctx->err.empty()) ctx->err = msg ? msg : "invalid batch";
      if (begun) drain_streams(ctx, st);
      return crc;
    }
  }
  run.have_heavy = !cut.chunks.empty() && cut.chunks[0].heavy;
  const float dbg_a = host_ms();
  if (int prc = prepare(false)) {
    if (cut.n_early) drain_streams(ctx, st);
    return prc;
  }
  run.cev.resize(cut.chunks.size(), ChunkEv{});
  const float dbg_b = host_ms();
  if (cut.use_lane)
    if (int lrc = launch_lane(run, n)) {
      drain_streams(ctx, st);
      return lrc;
    }

  // ---- plan (worker threads, chunk order) and launch (this thread, chunk order) ----
  int rc = SDF_OK;
  float dbg_c = 0.f;
  {
    // (batches of ordinary size are planned by this thread, a chunk ahead of the GPU: 25,000 tasks take under a
    // millisecond to plan and five to run; batches of many small tasks are planned on the context's parked threads)
    const int nthr = ctx->pool && cut.chunks.size() >= 3
                         ? (int)std::min<size_t>(ctx->pool->size(), cut.chunks.size())
                         : 0;
    ChunkPlanner planner(env, cut, run.plan, run.order, ctx->pool, nthr, cut.n_early);
    dbg_c = host_ms();
    for (size_t ci = cut.n_early; ci < cut.chunks.size() && rc == SDF_OK; ++ci) {
      planner.wait(ci);
      if (ci == 0) ctx->ms[4] = host_ms();
      const ChunkPlan &c = cut.chunks[ci];
      if (c.err) {
        ctx->err = c.err;
        rc = SDF_ERR_INVALID;
        break;
      }
      ctx->paired += c.paired;
      const float tw = host_ms();
      rc = launch_chunk(run, ci);
      if (dbg_plan_chunks)
        fprintf(stderr, "[chunk %zu: %zu tasks%s planned by %.2f ms, launched by %.2f ms; flags %.3f GB placed in a region of %.3f GB (the bound)]\n", ci,
                c.cnt, c.heavy ? " (heavy)" : "", tw, host_ms(), (double)c.dir_bytes * 1e-9,
                (double)(c.heavy ? cut.heavy_need : cut.region_need) * 1e-9);
    }
  }
  if (rc != SDF_OK) {  // earlier chunks are in flight and reference the context's buffers: let them finish
    drain_streams(ctx, st);
    return rc;
  }
  if (ctx->cfg.debug_plan)
    fprintf(stderr, "[plan: n=%zu cut %.2f ms, buffers %.2f ms, planner up %.2f ms, first chunk planned %.2f ms, all launched %.2f ms; chunks %zu heavy %zu]\n", n,
            dbg_a, dbg_b, dbg_c, ctx->ms[4], host_ms(), cut.chunks.size(), cut.n_heavy);
  return SDF_OK;
}

extern "C" int sdf_extz2_batch_device(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks,
                                      size_t n, const uint32_t *d_pool, uint32_t want,
                                      sdf_result *d_out, uint32_t *d_cig, size_t cigar_cap,
                                      size_t *cigar_used, void *stream_) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  const auto host_t0 = std::chrono::steady_clock::now();
  auto host_ms = [&] { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count(); };
  if (cigar_used) *cigar_used = 0;
  if (n == 0) return SDF_OK;
  if (!tasks || !d_out || n > 0x7fffffffu) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
  const bool dbg_plan_chunks = ctx->cfg.debug_plan != 0;
  // SDF_SPLIT_MIN=<tasks>: a batch of that many tasks or more starts in two parts.  Everything the GPU waits for before
  // its first launch is a pass over the caller's task array (40 bytes per task: 1.6-3 ms for a million tasks on eight
  // threads); the first part -- an eighth of the tasks (SDF_SPLIT_DIV) -- is cut, planned and launched on a second context
  // of this device, on a thread of its own, while this thread reads the rest.  The parts write disjoint ranges of the
  // result array; the CIGAR scan and the compaction at the end cover both.  Measured on the 1,000,000-task hg19 mixture:
  // 16.5 ms against 17.2 ms on one box, 17.7 against 15.4 on another -- within the box-to-box noise, and a second set of
  // streams and buffers: off by default, kept under test (tests/test_gpu_extz2.py).
  // By default (end of round 3) batches of 50,000 to 400,000 tasks of one size (see uniform_mid) on a process's only context
  // start that way with a first part of 8,192 tasks: their cut is a pass on ONE thread (1 ms for the 100,000 tasks of the headline batch) that the
  // first launch no longer waits for -- 1,110 against 1,088-1,090 Gcell/s, three runs each on one box.  Larger batches
  // start their heavy chunks early instead (cut_batch's two passes); SDF_SPLIT_MIN=0 turns the split off.
  const long long split_env = (long long)ctx->cfg.split_min;  // (-1: the default rule; 0: off)
  const int split_div = (int)ctx->cfg.split_div;
  // (only batches of tasks of one size, none of them long or small, by a sample of 256: 8,192 of them must keep the device
  // busy for the millisecond the cut of the rest takes, and a second context's streams next to the many small launches of a
  // batch of mixed lengths cost more than the early start brings -- mm8-like mixed bands, 100,000 tasks: 585 ms against 441;
  // hg19-shaped 250,000: 7.8 against 7.3)
  auto uniform_mid = [&] {
    const size_t step = n / 256;
    int mn = 0x7fffffff, mx = 0;
    double cells = 0;
    for (size_t i = 0; i < 256; ++i) {
      const sdf_task &t = tasks[i * step];
      const int d = t.qlen + t.tlen;
      mn = std::min(mn, d);
      mx = std::max(mx, d);
      const double full = (double)t.qlen * t.tlen;
      cells += t.w < 0 ? full : std::min(full, (2.0 * t.w + 1) * std::min(t.qlen, t.tlen));
    }
    return mn > 0 && mx <= 2 * mn && mx <= 4096 && cells >= 256 * 1e5;
  };
  const bool split_default = split_env < 0 && n >= 50000 && n < 400000 && g_live_contexts.load() == 1 && uniform_mid();
  const bool split_asked = split_env > 0 && n >= (size_t)split_env;
  BatchRun run, first;
  BatchRun *head = nullptr;
  size_t n_first = 0;
  if (ctx->pipeline && (split_default || split_asked) && !ctx->is_part) {
    if (!ctx->part_ctx) {
      sdf_config pc = ctx->cfg;  // (the parent's settings, whatever the environment says now)
      pc.workspace_gib = 0;
      pc.debug_plan = 0;
      ctx->part_ctx = sdf_create_cfg(ctx->device, ctx->ws_budget / 4, &pc);
      if (ctx->part_ctx) mark_internal_context(ctx->part_ctx);  // (a part context is not another user of the process's CPUs)
    }
    // (a batch of fewer than two blocks has no second part: SDF_SPLIT_MIN below 4,096 tasks would otherwise round the first
    // part up past the end of the batch -- ADVICE r3)
    size_t want_first = split_asked ? (n / split_div + SDF_CUT_BLOCK - 1) / SDF_CUT_BLOCK * SDF_CUT_BLOCK : 2 * SDF_CUT_BLOCK;
    if (n >= 2 * SDF_CUT_BLOCK && want_first + SDF_CUT_BLOCK > n) want_first = (n - SDF_CUT_BLOCK) / SDF_CUT_BLOCK * SDF_CUT_BLOCK;
    sdf_ctx *pc = n >= 2 * SDF_CUT_BLOCK && want_first >= SDF_CUT_BLOCK ? ctx->part_ctx : nullptr;
    if (pc) {
      n_first = want_first;
      // the part's stream starts where the caller's stream is
      if (ctx->part_ev == nullptr) (void)hipEventCreate(&ctx->part_ev);
      SDF_HIP(hipEventRecord(ctx->part_ev, st));
      SDF_HIP(hipStreamWaitEvent(pc->stream, ctx->part_ev, 0));
      head = &first;
    }
  }
  // (the first part on a thread of its own -- it plans on that thread alone, the planning threads are this part's --
  // while this thread reads the rest of the task array)
  int rc_first = SDF_OK;
  std::thread first_thread;
  if (head)
    first_thread = std::thread([&] {
      (void)hipSetDevice(ctx->device);
      rc_first = batch_part(ctx->part_ctx, sc, tasks, n_first, 0, d_pool, want, d_out, ctx->part_ctx->stream, first);
    });
  const int rc_main = batch_part(ctx, sc, tasks + n_first, n - n_first, n, d_pool, want, d_out + n_first, st, run);
  if (head) first_thread.join();
  if (rc_first != SDF_OK || rc_main != SDF_OK) {
    if (head) drain_streams(head->ctx, head->st);
    drain_streams(ctx, st);
    if (rc_main == SDF_OK) ctx->err = head->ctx->err;
    return rc_main != SDF_OK ? rc_main : rc_first;
  }
  int rc = finish_batch(run, head, n, d_out, d_cig, cigar_cap, cigar_used);
  if (rc != SDF_OK) {
    drain_streams(ctx, st);
    if (head) drain_streams(head->ctx, head->st);
    return rc;
  }
  if (head) {  // the call's statistics cover both parts
    ctx->launches += head->ctx->launches;
    ctx->paired += head->ctx->paired;
    ctx->lane_tasks += head->ctx->lane_tasks;
    ctx->reran += head->ctx->reran;
    ctx->ms[4] = head->ctx->ms[4];  // host time before the call's first launch
  }
  ctx->ms[5] = host_ms();
  if (dbg_plan_chunks) fprintf(stderr, "[batch finished by %.2f ms]\n", ctx->ms[5]);
  return SDF_OK;
}

__global__ void brief_results_kernel(const sdf_result *__restrict__ res, sdf_result_brief *__restrict__ out, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  sdf_result_brief b;
  b.cigar_off = res[k].cigar_off;
  b.n_cigar = (int32_t)res[k].n_cigar;
  b.matches = res[k].matches;
  out[k] = b;
}

// (view: the caller reads records and CIGAR words where the device's copies land -- the context's pinned staging, valid until
// the context's next call -- instead of receiving copies in arrays of its own)
struct ResultView {
  bool brief = true;
  const void *res = nullptr;
  const uint32_t *cig = nullptr;
};
static int batch_host_tail(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *t2, size_t n, size_t words, uint32_t want,
                           sdf_result *out, sdf_result_brief *brief, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used,
                           int nthr, std::chrono::steady_clock::time_point dbg0, std::chrono::steady_clock::time_point dbg1,
                           const char *what, ResultView *view = nullptr);

// The host-buffer call: sequences packed into pinned memory, one upload, the device-resident call, results and CIGARs back
// through pinned staging.  `brief`: 16-byte records instead of sdf_result (sdf_extz2_batch_brief).
static int batch_host(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, const uint8_t *seq_pool,
                      size_t pool_bytes, uint32_t want, sdf_result *out, sdf_result_brief *brief, uint32_t *cigar_pool,
                      size_t cigar_cap, size_t *cigar_used) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (cigar_used) *cigar_used = 0;
  if (n == 0) return SDF_OK;
  if (!tasks || (!out && !brief) || (!seq_pool && pool_bytes)) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  // pack every referenced sequence once (2-bit codes + N mask) and rewrite offsets to words
  const auto dbg0 = std::chrono::steady_clock::now();
  // (the task array with word offsets: kept by the context -- a fresh vector of 700,000 tasks is 34 MB of page faults per
  // call -- the offsets written here, the other fields copied by the packing threads below)
  std::vector<sdf_task> &t2 = ctx->host_tasks;
  if (t2.size() < n) t2.resize(n + n / 2);
  size_t words = 0;
  for (size_t k = 0; k < n; ++k) {
    const sdf_task &t = tasks[k];
    if (t.qlen < 0 || t.tlen < 0 || t.q_off < 0 || t.t_off < 0 ||
        (size_t)t.q_off + (size_t)t.qlen > pool_bytes || (size_t)t.t_off + (size_t)t.tlen > pool_bytes) {
      ctx->err = "task sequence range outside the pool";
      return SDF_ERR_INVALID;
    }
    t2[k].q_off = (int64_t)words;
    words += sdf_packed_words(t.qlen);
    t2[k].t_off = (int64_t)words;
    words += sdf_packed_words(t.tlen);
  }
  // packed straight into pinned memory (the upload is then one asynchronous DMA), on a few threads when the batch is
  // large; results and CIGARs come back through pinned staging too (a pageable hipMemcpy runs at ~2.5 GB/s here)
  SDF_HIP(ctx->host_pool.reserve(std::max<size_t>(words, 1) * 4));
  uint32_t *packed = (uint32_t *)ctx->host_pool.p;
  auto pack_range = [&](size_t lo, size_t hi) {
    for (size_t k = lo; k < hi; ++k) {
      const int64_t qo = t2[k].q_off, to = t2[k].t_off;
      t2[k] = tasks[k];
      t2[k].q_off = qo;
      t2[k].t_off = to;
      if (tasks[k].qlen > 0) sdf_pack_codes(seq_pool + tasks[k].q_off, tasks[k].qlen, packed + t2[k].q_off);
      if (tasks[k].tlen > 0) sdf_pack_codes(seq_pool + tasks[k].t_off, tasks[k].tlen, packed + t2[k].t_off);
    }
  };
  // (at most four when the process has several contexts -- the stage driver runs up to three of these calls at once next to
  // its own worker threads --, eight for a process's only context)
  const unsigned thr_cap = g_live_contexts.load() > 1 ? 4u : (unsigned)std::max(1, std::min(8, usable_cpus() / 2));
  const int nthr = words >= (1u << 18) ? (int)std::min<unsigned>(thr_cap, std::max(1u, std::thread::hardware_concurrency())) : 1;
  {
    // equal shares of words, not of tasks
    std::vector<size_t> cut(nthr + 1, n);
    cut[0] = 0;
    for (int q = 1; q < nthr; ++q) {
      const int64_t target = (int64_t)(words * (size_t)q / (size_t)nthr);
      size_t lo = cut[q - 1], hi = n;
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (t2[mid].q_off < target) lo = mid + 1; else hi = mid;
      }
      cut[q] = lo;
    }
    std::vector<std::thread> thr;
    for (int q = 1; q < nthr; ++q) thr.emplace_back(pack_range, cut[q], cut[q + 1]);
    pack_range(cut[0], cut[1]);
    for (auto &th : thr) th.join();
  }
  const auto dbg1 = std::chrono::steady_clock::now();
  SDF_HIP(ctx->h_pool.reserve(std::max<size_t>(words, 1) * 4));
  SDF_HIP(hipMemcpyAsync(ctx->h_pool.p, packed, std::max<size_t>(words, 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  return batch_host_tail(ctx, sc, t2.data(), n, words, want, out, brief, cigar_pool, cigar_cap, cigar_used, nthr, dbg0, dbg1, "sdf_extz2_batch");
}

// ... the rest of a host-buffer call once the packed sequences are (being) written to ctx->h_pool on ctx->stream: the
// device-resident call, results and CIGAR words back through pinned staging.
static int batch_host_tail(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *t2, size_t n, size_t words, uint32_t want,
                           sdf_result *out, sdf_result_brief *brief, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used,
                           int nthr, std::chrono::steady_clock::time_point dbg0, std::chrono::steady_clock::time_point dbg1,
                           const char *what, ResultView *view) {
  const bool dbg_t = ctx->cfg.debug_timing != 0;
  SDF_HIP(ctx->h_out.reserve(n * sizeof(sdf_result)));
  SDF_HIP(ctx->h_cig.reserve(std::max<size_t>(cigar_cap, 1) * 4));
  size_t used = 0;
  int rc = sdf_extz2_batch_device(ctx, sc, t2, n, (const uint32_t *)ctx->h_pool.p, want,
                                  (sdf_result *)ctx->h_out.p, (uint32_t *)ctx->h_cig.p, cigar_cap, &used,
                                  ctx->stream);
  if (cigar_used) *cigar_used = used;
  if (rc != SDF_OK) return rc;
  const auto dbg2 = std::chrono::steady_clock::now();
  const bool as_brief = view ? view->brief : brief != nullptr;
  const size_t out_bytes = n * (as_brief ? sizeof(sdf_result_brief) : sizeof(sdf_result)), cig_bytes = (used && (cigar_pool || view)) ? used * 4 : 0;
  SDF_HIP(ctx->host_out.reserve(out_bytes + cig_bytes + 64));
  uint8_t *stg = (uint8_t *)ctx->host_out.p;
  const void *d_res = ctx->h_out.p;
  if (as_brief) {
    SDF_HIP(ctx->h_brief.reserve(out_bytes));
    hipLaunchKernelGGL(brief_results_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const sdf_result *)ctx->h_out.p, (sdf_result_brief *)ctx->h_brief.p, (int)n);
    d_res = ctx->h_brief.p;
  }
  void *const out_any = brief ? (void *)brief : (void *)out;
  SDF_HIP(hipMemcpyAsync(stg, d_res, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (cig_bytes) SDF_HIP(hipMemcpyAsync(stg + out_bytes, ctx->h_cig.p, cig_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDF_HIP(hipStreamSynchronize(ctx->stream));
  if (view) {
    view->res = stg;
    view->cig = (const uint32_t *)(stg + out_bytes);
  } else {
    auto copy_range = [&](int q, int of) {
      const size_t a = out_bytes * (size_t)q / (size_t)of, b = out_bytes * (size_t)(q + 1) / (size_t)of;
      memcpy((uint8_t *)out_any + a, stg + a, b - a);
      const size_t c = cig_bytes * (size_t)q / (size_t)of, d = cig_bytes * (size_t)(q + 1) / (size_t)of;
      if (d > c) memcpy((uint8_t *)cigar_pool + c, stg + out_bytes + c, d - c);
    };
    const int nc = out_bytes + cig_bytes >= (8u << 20) ? std::min(nthr > 1 ? nthr : 4, 4) : 1;
    std::vector<std::thread> thr;
    for (int q = 1; q < nc; ++q) thr.emplace_back(copy_range, q, nc);
    copy_range(0, nc);
    for (auto &th : thr) th.join();
  }
  if (dbg_t) {
    const auto dbg3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    fprintf(stderr, "[%s n=%zu words=%zu cap=%zu used=%zu] pack %.1f ms, h2d+device %.1f ms (plan %.1f, dp %.1f, tb %.1f), d2h %.1f ms%s\n",
            what, n, words, cigar_cap, used, ms(dbg0, dbg1), ms(dbg1, dbg2), ctx->ms[4], ctx->ms[0], ctx->ms[1], ms(dbg2, dbg3),
            ctx->reran ? (", " + std::to_string(ctx->reran) + " tasks given up by a stripe wait and run again").c_str() : "");
  }
  return SDF_OK;
}

extern "C" int sdf_extz2_batch(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                               const uint8_t *seq_pool, size_t pool_bytes, uint32_t want,
                               sdf_result *out, uint32_t *cigar_pool, size_t cigar_cap,
                               size_t *cigar_used) {
  return batch_host(ctx, sc, tasks, n, seq_pool, pool_bytes, want, out, nullptr, cigar_pool, cigar_cap, cigar_used);
}

extern "C" int sdf_extz2_batch_brief(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                                     const uint8_t *seq_pool, size_t pool_bytes, sdf_result_brief *out,
                                     uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used) {
  return batch_host(ctx, sc, tasks, n, seq_pool, pool_bytes, SDF_WANT_CIGAR | SDF_WANT_SCORE, nullptr, out, cigar_pool, cigar_cap,
                    cigar_used);
}

// ---- resident sequences (include/sedef_hip.h; seq_pack.hip) --------------------------------------------------------
extern "C" char *sdf_pool_host(sdf_ctx *ctx, size_t bytes) {
  if (!ctx) return nullptr;
  ctx->err.clear();
  const auto t0 = std::chrono::steady_clock::now();
  const size_t had = ctx->host_chars.cap;
  const bool plain = ctx->cfg.pin_register < 2;  // (sdf_config: the pool crosses PCIe every super-batch -- see pin_register)
  if (hipSetDevice(ctx->device) != hipSuccess ||
      (plain ? ctx->host_chars.reserve_exact(std::max<size_t>(bytes, 64)) : ctx->host_chars.reserve_huge(std::max<size_t>(bytes, 64))) != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "cannot pin the character pool's staging";
    return nullptr;
  }
  (void)ctx->an_pool.reserve(bytes + 64);  // (its place in HBM with it: a first upload of 180 MB waited 8 ms for this)
  if (ctx->cfg.debug_timing && ctx->host_chars.cap != had)
    fprintf(stderr, "[sdf_pool_host %zu MiB %s in %.1f ms]\n", ctx->host_chars.cap >> 20, ctx->host_chars.registered ? "registered huge pages" : "hipHostMalloc",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return (char *)ctx->host_chars.p;
}

extern "C" int sdf_pool_upload(sdf_ctx *ctx, const char *chars, size_t bytes) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  ctx->pool_bytes = 0;
  if (!chars && bytes) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  ctx->an_pool.new_call();
  SDF_HIP(ctx->an_pool.reserve(bytes + 64));
  if (bytes) SDF_HIP(hipMemcpyAsync(ctx->an_pool.p, chars, bytes, hipMemcpyHostToDevice, ctx->stream));
  ctx->pool_bytes = bytes;
  return SDF_OK;
}

extern "C" size_t sdf_pool_bytes(const sdf_ctx *ctx) { return ctx ? ctx->pool_bytes : 0; }

static int batch_pairs(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, uint32_t want, sdf_result *out,
                       sdf_result_brief *brief, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used,
                       ResultView *view = nullptr) {
  using sdf::PackRec;
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (cigar_used) *cigar_used = 0;
  if (n == 0) return SDF_OK;
  if (!tasks || (!out && !brief && !view) || n > 0x3fffffffu) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  const auto dbg0 = std::chrono::steady_clock::now();
  // The task array with WORD offsets for the planner and the kernels, and one 32-byte record per task for the packing
  // kernel: where the task's characters are and where its packed words go.  Two passes over blocks of tasks on a few
  // threads (700,000 tasks a round in the chr1-sized stage run): the words of each block, then the records.
  std::vector<sdf_task> &t2 = ctx->host_tasks;
  if (t2.size() < n) t2.resize(n + n / 2);
  SDF_HIP(ctx->host_pool.reserve(n * sizeof(PackRec)));
  PackRec *recs = (PackRec *)ctx->host_pool.p;
  const size_t block = 32768, nb = (n + block - 1) / block, pool_bytes = ctx->pool_bytes;
  std::vector<size_t> bwords(nb + 1, 0), bcap(nb, 0);
  std::atomic<bool> bad(false);
  const unsigned thr_cap = g_live_contexts.load() > 1 ? 4u : (unsigned)std::max(1, std::min(8, usable_cpus() / 2));
  const int nthr = (int)std::min<size_t>(nb, thr_cap);
  auto on_blocks = [&](const std::function<void(size_t)> &f) {
    std::atomic<size_t> next(0);
    auto work = [&] {
      for (size_t b = next.fetch_add(1); b < nb; b = next.fetch_add(1)) f(b);
    };
    std::vector<std::thread> thr;
    for (int q = 1; q < nthr; ++q) thr.emplace_back(work);
    work();
    for (auto &th : thr) th.join();
  };
  on_blocks([&](size_t b) {
    size_t w = 0, c = 0;
    for (size_t k = b * block; k < std::min(n, (b + 1) * block); ++k) {
      const sdf_task &t = tasks[k];
      if (t.qlen < 0 || t.tlen < 0 || t.q_off < 0 || t.t_off < 0 || (size_t)t.q_off + (size_t)t.qlen > pool_bytes ||
          (size_t)t.t_off + (size_t)t.tlen > pool_bytes)
        bad.store(true);
      w += sdf_packed_words(t.qlen) + sdf_packed_words(t.tlen);
      c += (size_t)t.qlen + (size_t)t.tlen + 2;
    }
    bwords[b + 1] = w;
    bcap[b] = c;
  });
  if (view) {  // (no caller's pool to fit: the staging takes the words there are; the device's bound is the worst case)
    cigar_cap = 16;
    for (size_t b = 0; b < nb; ++b) cigar_cap += bcap[b];
  }
  if (bad.load()) {
    ctx->err = "task sequence range outside the resident pool (sdf_pool_upload)";
    return SDF_ERR_INVALID;
  }
  for (size_t b = 0; b < nb; ++b) bwords[b + 1] += bwords[b];
  const size_t words = bwords[nb];
  on_blocks([&](size_t b) {
    size_t w = bwords[b];
    for (size_t k = b * block; k < std::min(n, (b + 1) * block); ++k) {
      const sdf_task &t = tasks[k];
      PackRec &r = recs[k];
      r.q_byte = t.q_off;
      r.t_byte = t.t_off;
      r.q_word = (int64_t)w;
      r.qlen = t.qlen;
      r.tlen = t.tlen;
      t2[k] = t;
      t2[k].q_off = (int64_t)w;
      w += sdf_packed_words(t.qlen);
      t2[k].t_off = (int64_t)w;
      w += sdf_packed_words(t.tlen);
    }
  });
  const auto dbg1 = std::chrono::steady_clock::now();
  for (DevBuf *b : {&ctx->pk_recs}) b->new_call();
  SDF_HIP(ctx->h_pool.reserve(std::max<size_t>(words, 1) * 4));
  SDF_HIP(ctx->pk_recs.reserve(n * sizeof(PackRec)));
  SDF_HIP(hipMemcpyAsync(ctx->pk_recs.p, recs, n * sizeof(PackRec), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(sdf::pack_chars_kernel, dim3((unsigned)((2 * n + 15) / 16)), dim3(256), 0, ctx->stream,
                     (const PackRec *)ctx->pk_recs.p, (long long)(2 * n), (const char *)ctx->an_pool.p, (uint32_t *)ctx->h_pool.p);
  return batch_host_tail(ctx, sc, t2.data(), n, words, want, out, brief, cigar_pool, cigar_cap, cigar_used, nthr, dbg0, dbg1,
                         "sdf_extz2_batch_pairs", view);
}

extern "C" int sdf_extz2_batch_pairs_view(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                                          const sdf_result_brief **out, const uint32_t **cigar_pool, size_t *cigar_used) {
  if (!out || !cigar_pool) return SDF_ERR_INVALID;
  *out = nullptr;
  *cigar_pool = nullptr;
  ResultView v;
  const int rc = batch_pairs(ctx, sc, tasks, n, SDF_WANT_CIGAR | SDF_WANT_SCORE, nullptr, nullptr, nullptr, 0, cigar_used, &v);
  if (rc == SDF_OK) {
    *out = (const sdf_result_brief *)v.res;
    *cigar_pool = v.cig;
  }
  return rc;
}

extern "C" int sdf_extz2_batch_pairs(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, sdf_result_brief *out,
                                     uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used) {
  return batch_pairs(ctx, sc, tasks, n, SDF_WANT_CIGAR | SDF_WANT_SCORE, nullptr, out, cigar_pool, cigar_cap, cigar_used);
}

extern "C" int sdf_extz2_batch_pairs_full(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n, uint32_t want,
                                          sdf_result *out, uint32_t *cigar_pool, size_t cigar_cap, size_t *cigar_used) {
  return batch_pairs(ctx, sc, tasks, n, want, out, nullptr, cigar_pool, cigar_cap, cigar_used);
}

// Buffers sized once (include/sedef_hip.h).  The bounds per task are the planner's: a launch-order entry per task and
// stripe / block of columns, a CIGAR staging slot of qlen + tlen + 2 words.
extern "C" int sdf_reserve(sdf_ctx *ctx, size_t max_tasks, size_t max_bases, size_t workspace_bytes, uint32_t flags) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  SDF_HIP(hipSetDevice(ctx->device));
  const auto rt0 = std::chrono::steady_clock::now();
  auto lap = [&, last = rt0](const char *what) mutable {  // (SDF_DEBUG_TIMING: the sections that took more than 20 ms)
    const auto t = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t - last).count();
    if (ctx->cfg.debug_timing && ms > 20) fprintf(stderr, "[sdf_reserve: %s %.1f ms]\n", what, ms);
    last = t;
  };
  const size_t n = std::max<size_t>(max_tasks, 1);
  const size_t words = max_bases / 16 + max_bases / 32 + 4 * n + 16;  // (packed sequences: two roundings per sequence)
  const size_t cig_words = max_bases + 2 * n + 16;
  const size_t nord = 3 * n + max_bases / 16 + 1024;
  // pinned staging (registered huge pages unless sdf_config.pin_register says otherwise: sdf_ctx.h, HostBuf::reserve_huge)
  const bool reg_small = ctx->cfg.pin_register >= 1, reg_big = ctx->cfg.pin_register >= 2;
  SDF_HIP(ctx->host_pool.reserve_pinned(reg_small, std::max(words * 4, n * sizeof(sdf::PackRec))));  // (packed sequences, or a record per task of sdf_extz2_batch_pairs)
  SDF_HIP(ctx->pk_recs.reserve_exact(n * sizeof(sdf::PackRec)));
  SDF_HIP(ctx->host_plan.reserve_pinned(reg_small, n * sizeof(PlanTask)));
  SDF_HIP(ctx->host_order.reserve_pinned(reg_small, nord * sizeof(int32_t)));
  SDF_HIP(ctx->host_lane.reserve_pinned(reg_small, n * sizeof(LaneRec)));
  // (results + CIGAR words: a quarter of the CIGAR bound -- the stage's rounds fill a tenth of it)
  SDF_HIP(ctx->host_out.reserve_pinned(reg_small, n * ((flags & SDF_RESERVE_BRIEF) ? sizeof(sdf_result_brief) : sizeof(sdf_result)) +
                                      cig_words / 4 * 4 + 64));
  if (ctx->host_tasks.size() < n) ctx->host_tasks.resize(n);
  lap("pinned staging");
  // device
  SDF_HIP(ctx->h_pool.reserve_exact(words * 4));
  SDF_HIP(ctx->h_out.reserve_exact(n * sizeof(sdf_result)));
  SDF_HIP(ctx->h_brief.reserve_exact(n * sizeof(sdf_result_brief)));
  SDF_HIP(ctx->h_cig.reserve_exact(cig_words * 4));
  SDF_HIP(ctx->stage_ws.reserve_exact(cig_words * 4));
  SDF_HIP(ctx->plan_buf.reserve_exact(2 * n * sizeof(PlanTask)));  // (host-planned records, the lane tasks' behind them)
  SDF_HIP(ctx->order_buf.reserve_exact(nord * sizeof(int32_t)));
  SDF_HIP(ctx->misc_buf.reserve_exact(SDF_MISC_PARTS * 8 + ((n + 1023) / 1024 + 1) * 8));
  SDF_HIP(ctx->claim_buf.reserve_exact(kClaimSets * 8 * sizeof(unsigned)));
  SDF_HIP(ctx->ln_recs.reserve_exact(n * sizeof(LaneRec)));
  SDF_HIP(ctx->ln_keys.reserve_exact(n * 8));
  SDF_HIP(ctx->ln_vals.reserve_exact(n * 8));
  SDF_HIP(ctx->ln_sizes.reserve_exact(n * 32 + 64));
  SDF_HIP(ctx->ln_bins.reserve_exact((size_t)kLaneBins * (4 + 4 + 4 + 8 + 8) + (size_t)(kLaneBins / kLaneScanBlock) * 24 + 256));
  lap("device buffers");
  {  // (the library sort / scan of the lane tasks' planning: sdf_launch.hip, launch_lane)
    size_t t_sort = 0, t_scan = 0;
    SDF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, t_sort, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                               (uint32_t *)nullptr, (int)n, 0, 20, ctx->stream));
    SDF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t_scan, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)n,
                                             ctx->stream));
    SDF_HIP(ctx->ln_tmp.reserve_exact(std::max(t_sort, t_scan) + 256));
  }
  // the streams the pipeline would create the first time it wants them (a stream is a hardware queue: 7-15 ms each to set
  // up -- the stage's first two rounds spent 35 ms on five of them)
  lap("sort / scan scratch");
  if (flags & SDF_RESERVE_FEW_STREAMS) ctx->aux_limit = 0;
  if (ctx->pipeline) {
    for (hipStream_t *q : {&ctx->lane_stream, &ctx->aux_stream[0], &ctx->aux_stream[1], &ctx->aux_stream[2], &ctx->aux_stream[3]}) {
      if (q != &ctx->lane_stream && (size_t)(q - &ctx->aux_stream[0]) >= ctx->aux_limit) continue;
      if (!*q && (q == &ctx->lane_stream ? create_lane_stream(ctx, q) : hipStreamCreateWithFlags(q, hipStreamNonBlocking)) != hipSuccess) {
        (void)hipGetLastError();
        *q = nullptr;
      }
    }
  }
  lap("pipeline streams");
  if (workspace_bytes) {
    const size_t ws = std::min(workspace_bytes, ctx->ws_budget);
    if (ctx->dir_ws.reserve_exact(ws) != hipSuccess) {
      (void)hipGetLastError();
      ctx->err = "cannot allocate the direction-matrix workspace";
      return SDF_ERR_NOMEM;
    }
  }
  lap("direction-flag workspace");
  if (flags & SDF_RESERVE_ANCHORS) SDF_HIP(ctx->host_an.reserve_pinned(reg_big, (size_t)48 << 20));
  lap("pinned anchors staging");
  if (flags & SDF_RESERVE_ANCHORS) {  // two copies of a short sequence: a handful of anchors through every kernel of the path
    char seq[192];
    uint32_t x = 12345u;
    for (int i = 0; i < 96; ++i) {
      x = x * 1664525u + 1013904223u;
      seq[i] = seq[96 + i] = "ACGT"[x >> 30];
    }
    sdf_anchor_pair pr;
    memset(&pr, 0, sizeof(pr));
    pr.q_off = 0;
    pr.r_off = 96;
    pr.qlen = pr.rlen = 96;
    sdf_anchor out[256];
    int64_t off[2];
    size_t used = 0;
    (void)sdf_anchors_batch(ctx, &pr, 1, seq, sizeof(seq), 11, out, 256, off, &used);
    ctx->err.clear();
  }
  lap("anchors warm-up call");
  {  // the stream's first asynchronous copy in each direction costs its caller ~8 ms (the runtime sets its copy path up): here,
     // not in front of a super-batch's upload and its anchors' way back (profiles/r06_stage_timeline.txt)
    const size_t probe = std::min<size_t>({(size_t)1 << 20, ctx->host_pool.cap, ctx->h_pool.cap, ctx->host_out.cap, ctx->h_out.cap});
    if (probe) {
      SDF_HIP(hipMemcpyAsync(ctx->h_pool.p, ctx->host_pool.p, probe, hipMemcpyHostToDevice, ctx->stream));
      // (device to host: a copy of the size the rounds' results have -- a small one does not take the path a 17 MB one takes)
      const size_t back = std::min<size_t>({(size_t)32 << 20, ctx->host_out.cap, ctx->h_out.cap});
      SDF_HIP(hipMemcpyAsync(ctx->host_out.p, ctx->h_out.p, back, hipMemcpyDeviceToHost, ctx->stream));
      SDF_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  lap("first asynchronous copies");
  return SDF_OK;
}

// Debug: wavefronts started per (XCD, shader engine, CU, SIMD) since the last call, 4096 counters indexed
// xcd << 9 | se << 6 | cu << 2 | simd (the chained strips note theirs: how evenly the dispatcher spreads a launch).
extern "C" int sdf_debug_placement(sdf_ctx *ctx, uint32_t *out) {
  if (!ctx) return SDF_ERR_INVALID;
  SDF_HIP(hipSetDevice(ctx->device));
  static unsigned *buf = nullptr;
  if (!buf) {
    SDF_HIP(hipMalloc(&buf, 4096 * sizeof(unsigned)));
    SDF_HIP(hipMemset(buf, 0, 4096 * sizeof(unsigned)));
    SDF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sdf::g_place), &buf, sizeof(buf)));
  }
  SDF_HIP(hipDeviceSynchronize());
  if (out) {
    SDF_HIP(hipMemcpy(out, buf, 4096 * sizeof(unsigned), hipMemcpyDeviceToHost));
    SDF_HIP(hipMemset(buf, 0, 4096 * sizeof(unsigned)));
  }
  return SDF_OK;
}

extern "C" size_t sdf_device_bytes(const sdf_ctx *ctx) {
  if (!ctx) return 0;
  size_t sum = 0;
  for (const DevBuf *b : {&ctx->dir_ws, &ctx->stage_ws, &ctx->plan_buf, &ctx->order_buf, &ctx->misc_buf, &ctx->gstate_buf, &ctx->claim_buf,
                          &ctx->h_pool, &ctx->h_out, &ctx->h_brief, &ctx->h_cig, &ctx->rr_out, &ctx->rr_cig, &ctx->rr_map, &ctx->ln_recs,
                          &ctx->ln_keys, &ctx->ln_vals, &ctx->ln_sizes, &ctx->ln_tmp, &ctx->pk_recs,
                          // the anchors / chaining / stats entry points
                          &ctx->an_pool, &ctx->an_pairs, &ctx->an_keys, &ctx->an_keys2, &ctx->an_q, &ctx->an_off, &ctx->an_flag, &ctx->an_pos,
                          &ctx->an_cand, &ctx->an_out, &ctx->an_tmp, &ctx->an_outoff, &ctx->ch_an, &ctx->ch_off, &ctx->ch_wsoff, &ctx->ch_work,
                          &ctx->ch_path, &ctx->ch_bounds, &ctx->ch_nb, &ctx->ch_which, &ctx->st_tasks, &ctx->st_pool, &ctx->st_cig, &ctx->st_out})
    sum += b->held_bytes();
  if (ctx->part_ctx) sum += sdf_device_bytes(ctx->part_ctx);
  if (ctx->rerun_ctx) sum += sdf_device_bytes(ctx->rerun_ctx);
  return sum;
}

// ---- seed anchors (reference: src/chain.cc:24-101) ---------------------------------------------------
static int anchors_range(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *d_pool, int kmer, int pos_bits,
                         sdf_anchor *out, size_t out_cap, int64_t *out_off, size_t *out_used, hipStream_t st) {
  using namespace sdf;
  const auto lt0 = std::chrono::steady_clock::now();
  auto lap = [&, last = lt0](const char *what) mutable {  // (SDF_DEBUG_TIMING: host milliseconds of the call's sections)
    if (!ctx->cfg.debug_timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[anchors_range: %s %.2f ms]\n", what, std::chrono::duration<double, std::milli>(t - last).count());
    last = t;
  };
  int pair_bits = 1;
  while (((size_t)1 << pair_bits) <= n) ++pair_bits;  // (strictly more than n - 1 needs: the all-ones pair field is the invalid keys' alone)
  const int key_bits = std::min(64, pair_bits + 2 * kmer + pos_bits);  // (the sort looks at the bits in use only)
  std::vector<AnchorPairDev> hp(n);
  long long nrk = 0, nqk = 0;
  for (size_t i = 0; i < n; i++) {
    AnchorPairDev &d = hp[i];
    d.q_off = pairs[i].q_off;
    d.r_off = pairs[i].r_off;
    d.qlen = pairs[i].qlen;
    d.rlen = pairs[i].rlen;
    d.same_chr = pairs[i].same_chr;
    d.delta = pairs[i].delta;
    d.rk_start = nrk;
    d.qk_start = nqk;
    nrk += std::max(0, d.rlen - kmer + 1);
    nqk += std::max(0, d.qlen - kmer + 1);
  }
  for (size_t i = 0; i <= n; i++) out_off[i] = 0;
  *out_used = 0;
  if (nrk == 0 || nqk == 0) return SDF_OK;
  SDF_HIP(ctx->an_pairs.reserve(n * sizeof(AnchorPairDev)));
  SDF_HIP(ctx->an_keys.reserve((size_t)nrk * 8));
  SDF_HIP(ctx->an_keys2.reserve((size_t)nrk * 8));
  SDF_HIP(ctx->an_q.reserve((size_t)nqk * 16));
  SDF_HIP(ctx->an_off.reserve((size_t)(nqk + 1) * 8));
  SDF_HIP(ctx->an_outoff.reserve((n + 1) * 8));
  AnchorPairDev *d_pairs = (AnchorPairDev *)ctx->an_pairs.p;
  unsigned long long *d_keys = (unsigned long long *)ctx->an_keys.p, *d_keys2 = (unsigned long long *)ctx->an_keys2.p;
  uint32_t *d_qlo = (uint32_t *)ctx->an_q.p, *d_qcnt = d_qlo + nqk, *d_qeff = d_qcnt + nqk, *d_qpair = d_qeff + nqk;
  unsigned long long *d_off = (unsigned long long *)ctx->an_off.p;
  lap("pair records, buffers");
  SDF_HIP(hipMemcpyAsync(d_pairs, hp.data(), n * sizeof(AnchorPairDev), hipMemcpyHostToDevice, st));
  const dim3 grid(32, (unsigned)std::min<size_t>(n, 65535), (unsigned)((n + 65534) / 65535));
  hipLaunchKernelGGL(ref_keys_kernel, grid, dim3(256), 0, st, d_pairs, (int)n, d_pool, kmer, pos_bits, d_keys);
  size_t tmp_bytes = 0;
  SDF_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys, d_keys2, (int)nrk, pos_bits, key_bits, st));
  size_t scan_bytes = 0;
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (unsigned long long *)nullptr,
                                           (int)(nqk + 1), st));
  SDF_HIP(ctx->an_tmp.reserve(std::max(tmp_bytes, scan_bytes) + 256));
  // (the keys are written in ascending position inside each pair and the sort is stable: the position bits need no pass)
  SDF_HIP(hipcub::DeviceRadixSort::SortKeys(ctx->an_tmp.p, tmp_bytes, d_keys, d_keys2, (int)nrk, pos_bits, key_bits, st));
  hipLaunchKernelGGL(query_lookup_kernel, grid, dim3(256), 0, st, d_pairs, (int)n, d_pool, kmer, pos_bits, d_keys2, nrk, d_qlo,
                     d_qcnt, d_qeff, d_qpair);
  // exclusive scan over nqk+1 entries (the extra input element is ignored by the exclusive form)
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->an_tmp.p, scan_bytes, d_qeff, d_off, (int)(nqk + 1), st));
  unsigned long long ncand = 0;
  SDF_HIP(hipMemcpyAsync(&ncand, d_off + nqk, 8, hipMemcpyDeviceToHost, st));
  lap("keys, sort, lookup, scan enqueued");
  SDF_HIP(hipStreamSynchronize(st));
  lap("... done on the device");
  if (ncand == 0) return SDF_OK;
  if (ncand > (1ull << 30)) {
    ctx->err = "anchor candidates exceed 2^30 in one batch";
    return SDF_ERR_NOMEM;
  }
  SDF_HIP(ctx->an_flag.reserve((size_t)(ncand + 1) * 4));
  SDF_HIP(ctx->an_pos.reserve((size_t)(ncand + 1) * 8));
  SDF_HIP(ctx->an_cand.reserve((size_t)ncand * sizeof(CandOut)));
  uint32_t *d_flag = (uint32_t *)ctx->an_flag.p;
  unsigned long long *d_pos = (unsigned long long *)ctx->an_pos.p;
  CandOut *d_cand = (CandOut *)ctx->an_cand.p;
  const unsigned nb = (unsigned)((ncand + 255) / 256);
  hipLaunchKernelGGL(candidates_kernel, dim3(nb), dim3(256), 0, st, d_pairs, d_pool, kmer, d_keys2, d_qlo, d_qcnt, d_off,
                     d_qpair, nqk, (long long)ncand, d_flag, d_cand, pos_bits);
  SDF_HIP(hipMemsetAsync(d_flag + ncand, 0, 4, st));
  size_t scan2 = 0;
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan2, d_flag, d_pos, (int)(ncand + 1), st));
  SDF_HIP(ctx->an_tmp.reserve(scan2 + 256));
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->an_tmp.p, scan2, d_flag, d_pos, (int)(ncand + 1), st));
  unsigned long long total = 0;
  SDF_HIP(hipMemcpyAsync(&total, d_pos + ncand, 8, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
  lap("candidates + scan");
  *out_used = (size_t)total;
  long long *d_outoff = (long long *)ctx->an_outoff.p;
  hipLaunchKernelGGL(anchor_offsets_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, d_pairs, (int)n, d_off,
                     d_pos, (long long)ncand, total, nqk, d_outoff);
  SDF_HIP(hipMemcpyAsync(out_off, d_outoff, (n + 1) * 8, hipMemcpyDeviceToHost, st));
  if (total > out_cap) {
    SDF_HIP(hipStreamSynchronize(st));
    ctx->err = "anchor output buffer too small";
    return SDF_ERR_CIGAR_OVERFLOW;
  }
  if (total) {
    const size_t bytes = (size_t)total * sizeof(sdf_anchor);
    static_assert(sizeof(CandOut) == sizeof(sdf_anchor), "the compaction writes anchors as they go out");
    const bool out_is_pinned = (const uint8_t *)out >= (const uint8_t *)ctx->host_an.p &&
                               (const uint8_t *)out + bytes <= (const uint8_t *)ctx->host_an.p + ctx->host_an.cap;
    // (sdf_anchors_batch_view: the caller reads the pinned staging itself, and the compaction kernel WRITES it there -- sixteen
    // bytes a lane, coalesced, over PCIe; an asynchronous device-to-host copy of the same 34 MB behind the kernel cost its
    // caller 7-8 ms to enqueue)
    if (!out_is_pinned) SDF_HIP(ctx->an_out.reserve((size_t)total * sizeof(CandOut)));
    hipLaunchKernelGGL(anchors_compact_kernel, dim3(nb), dim3(256), 0, st, d_flag, d_pos, d_cand, (long long)ncand,
                       out_is_pinned ? (CandOut *)out : (CandOut *)ctx->an_out.p, total);
    if (out_is_pinned) {
    } else if (bytes >= ((size_t)1 << 20) && bytes <= ctx->host_an.cap) {  // through pinned staging, copied out on a few threads
      SDF_HIP(hipMemcpyAsync(ctx->host_an.p, ctx->an_out.p, bytes, hipMemcpyDeviceToHost, st));
      SDF_HIP(hipStreamSynchronize(st));
      const int nthr = 4;
      std::vector<std::thread> thr;
      auto part = [&](int q) {
        const size_t a = bytes * (size_t)q / nthr, b = bytes * (size_t)(q + 1) / nthr;
        memcpy((uint8_t *)out + a, (const uint8_t *)ctx->host_an.p + a, b - a);
      };
      for (int q = 1; q < nthr; ++q) thr.emplace_back(part, q);
      part(0);
      for (auto &t : thr) t.join();
    } else {
      SDF_HIP(hipMemcpyAsync(out, ctx->an_out.p, bytes, hipMemcpyDeviceToHost, st));
    }
  }
  SDF_HIP(hipStreamSynchronize(st));
  lap("compaction + anchors to the host");
  SDF_HIP(hipGetLastError());
  return SDF_OK;
}

extern "C" int sdf_anchors_batch(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *seq_pool,
                                 size_t pool_bytes, int kmer, sdf_anchor *out, size_t out_cap, int64_t *out_off,
                                 size_t *out_used) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (out_used) *out_used = 0;
  if (!pairs || !out_off || !out_used) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  const bool resident = !seq_pool && pool_bytes;  // (the characters sdf_pool_upload left in HBM)
  if (resident && pool_bytes > ctx->pool_bytes) {
    ctx->err = "the resident pool (sdf_pool_upload) is shorter than pool_bytes";
    return SDF_ERR_INVALID;
  }
  if (kmer < 1 || kmer > 15) {  // (the reference's hash is the 2-bit code of the k-mer in 32 bits, src/chain.cc:30-35)
    ctx->err = "GPU anchors implement k-mer sizes up to 15";
    return SDF_ERR_UNSUPPORTED;
  }
  int32_t rmax = 1;
  for (size_t i = 0; i < n; i++) {
    const sdf_anchor_pair &p = pairs[i];
    if (p.qlen < 0 || p.rlen < 0) {
      ctx->err = "negative sequence length";
      return SDF_ERR_INVALID;
    }
    rmax = std::max(rmax, p.rlen);
    if (p.q_off < 0 || p.r_off < 0 || (size_t)p.q_off + p.qlen > pool_bytes || (size_t)p.r_off + p.rlen > pool_bytes) {
      ctx->err = "pair sequence range outside the pool";
      return SDF_ERR_INVALID;
    }
  }
  SDF_HIP(hipSetDevice(ctx->device));
  if (n == 0) {
    out_off[0] = 0;
    return SDF_OK;
  }
  const bool dbg_t = ctx->cfg.debug_timing != 0;
  const auto dbg0 = std::chrono::steady_clock::now();
  if (!resident) {  // (the pool stays where it is after the call: sdf_extz2_batch_pairs may name ranges of it)
    ctx->pool_bytes = 0;
    SDF_HIP(ctx->an_pool.reserve(pool_bytes + 64));
    SDF_HIP(hipMemcpyAsync(ctx->an_pool.p, seq_pool, pool_bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->pool_bytes = pool_bytes;
  }
  if (dbg_t) SDF_HIP(hipStreamSynchronize(ctx->stream));
  const auto dbg1 = std::chrono::steady_clock::now();
  // Key = pair | hash (2k bits) | position: the pairs are run in ranges that fit the bits the other two fields leave (k = 11
  // and references of up to 100 kb: 33 million pairs a range; k = 15 and 5 Mb: 2,048) -- and whose k-mers fit 32-bit indices.
  int pos_bits = 1;
  while (pos_bits < 31 && ((int64_t)1 << pos_bits) < (int64_t)rmax) ++pos_bits;
  const int pair_bits = std::min(30, 64 - 2 * kmer - pos_bits);
  const size_t range_max = ((size_t)1 << pair_bits) - 1;  // (a range's pair field never reaches all ones: anchors_range)
  int rc = SDF_OK;
  size_t used_total = 0;
  out_off[0] = 0;
  for (size_t s = 0; s < n && rc == SDF_OK;) {
    size_t e = s;
    int64_t nrk = 0, nqk = 0;
    while (e < n && e - s < range_max) {
      const int64_t a = std::max(0, pairs[e].rlen - kmer + 1), b = std::max(0, pairs[e].qlen - kmer + 1);
      if (e > s && (nrk + a > 0x7fffff00ll || nqk + b > 0x7fffff00ll)) break;
      nrk += a, nqk += b;
      ++e;
    }
    if (nrk > 0x7fffff00ll || nqk > 0x7fffff00ll) {
      ctx->err = "a pair of sequences of 2 Gb or more";
      return SDF_ERR_UNSUPPORTED;
    }
    size_t used = 0;
    const int64_t first = out_off[s];
    rc = anchors_range(ctx, pairs + s, e - s, (const char *)ctx->an_pool.p, kmer, pos_bits, out ? out + used_total : nullptr,
                       out_cap > used_total ? out_cap - used_total : 0, out_off + s, &used, ctx->stream);
    for (size_t i = s; i <= e; i++) out_off[i] += first;  // (the range's offsets start at 0)
    if (rc == SDF_ERR_CIGAR_OVERFLOW) {  // the caller wants the size needed: count the remaining ranges too
      size_t more = 0;
      for (size_t s2 = e; s2 < n;) {
        size_t e2 = std::min(n, s2 + range_max), u2 = 0;
        std::vector<int64_t> tmp_off(e2 - s2 + 1);
        (void)anchors_range(ctx, pairs + s2, e2 - s2, (const char *)ctx->an_pool.p, kmer, pos_bits, nullptr, 0, tmp_off.data(), &u2,
                            ctx->stream);
        more += u2;
        s2 = e2;
      }
      used_total += used + more;
      break;
    }
    used_total += used;
    s = e;
  }
  *out_used = used_total;
  if (dbg_t)
    fprintf(stderr, "[sdf_anchors_batch n=%zu pool=%zu anchors=%zu] upload %.1f ms, rest %.1f ms\n", n, pool_bytes, *out_used,
            std::chrono::duration<double, std::milli>(dbg1 - dbg0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg1).count());
  return rc;
}

extern "C" int sdf_anchors_batch_view(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *seq_pool, size_t pool_bytes,
                                      int kmer, const sdf_anchor **out, int64_t *out_off, size_t *out_used) {
  if (!ctx || !out) return SDF_ERR_INVALID;
  *out = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || ctx->host_an.reserve_pinned(ctx->cfg.pin_register >= 2, (size_t)48 << 20) != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "cannot pin the anchors' staging";
    return SDF_ERR_NOMEM;
  }
  int rc = sdf_anchors_batch(ctx, pairs, n, seq_pool, pool_bytes, kmer, (sdf_anchor *)ctx->host_an.p, ctx->host_an.cap / sizeof(sdf_anchor),
                             out_off, out_used);
  if (rc == SDF_ERR_CIGAR_OVERFLOW) {  // more anchors than the staging holds: once more with room for all of them
    if (ctx->host_an.reserve_pinned(ctx->cfg.pin_register >= 2, (*out_used + 1024) * sizeof(sdf_anchor)) != hipSuccess) {
      (void)hipGetLastError();
      ctx->err = "cannot pin the anchors' staging";
      return SDF_ERR_NOMEM;
    }
    // (the characters are resident since the first attempt)
    rc = sdf_anchors_batch(ctx, pairs, n, nullptr, pool_bytes, kmer, (sdf_anchor *)ctx->host_an.p, ctx->host_an.cap / sizeof(sdf_anchor),
                           out_off, out_used);
  }
  if (rc == SDF_OK) *out = (const sdf_anchor *)ctx->host_an.p;
  return rc;
}

// ... of MORE pairs of the resident pool, written behind the first `keep` anchors of the staging (which stay where they are: a
// caller that is still reading them -- the stage driver chains the first half of a super-batch while the device finds the
// anchors of the second -- is not disturbed).  No growth: SDF_ERR_CIGAR_OVERFLOW when the staging has no room for them.
extern "C" int sdf_anchors_batch_more(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, size_t pool_bytes, int kmer, size_t keep,
                                      const sdf_anchor **out, int64_t *out_off, size_t *out_used) {
  if (!ctx || !out) return SDF_ERR_INVALID;
  *out = nullptr;
  const size_t cap = ctx->host_an.cap / sizeof(sdf_anchor);
  if (!ctx->host_an.p || keep > cap || !pool_bytes) {
    ctx->err = "sdf_anchors_batch_more follows sdf_anchors_batch_view on a resident pool";
    return SDF_ERR_INVALID;
  }
  sdf_anchor *at = (sdf_anchor *)ctx->host_an.p + keep;
  const int rc = sdf_anchors_batch(ctx, pairs, n, nullptr, pool_bytes, kmer, at, cap - keep, out_off, out_used);
  if (rc == SDF_OK) *out = at;
  return rc;
}

// ---- anchor chaining (reference: src/chain.cc:103-199) ---------------------------------------------------
extern "C" int sdf_chain_batch(sdf_ctx *ctx, const sdf_anchor *anchors, const int64_t *off, size_t n, int max_chain_gap,
                               int match_chain_score, int32_t *path, int32_t *bounds, int32_t *nbound) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (!off || !bounds || !nbound || n >= (1u << 24)) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  if (n == 0) return SDF_OK;
  // Round 4: a pair whose arrays fit the LDS of a workgroup is swept by ONE WAVEFRONT with everything in LDS
  // (chain_wave_kernel: launch classes by LDS size, the pairs of most anchors first); the others keep the thread-per-pair
  // kernel with its scratch in HBM.  SDF_CHAIN_THREADS=1: every pair on the latter (tests).
  const bool threads_only = ctx->cfg.chain_threads_only != 0;
  // (classes of up to 32 KiB, ~400 anchors, whatever their number; up to the device's LDS per workgroup when they are FEW: a wavefront
  // sweeps an anchor in ~14 us where a thread chasing nodes in HBM takes ~85 -- the launch is its largest pair --, but two
  // such workgroups fit a CU: 8,192 pairs of ~700 anchors take 150 ms that way against 59 ms with every pair in flight on
  // the thread-per-pair kernel; profiles/r04_chain_bench.txt)
  const size_t caps[6] = {2048, 4096, 8192, 16384, 32768, (size_t)std::max(ctx->max_dyn_lds, 65536)};
  std::vector<int32_t> cls[7];  // [6]: thread-per-pair
  std::vector<int64_t> ws_off(n + 1);
  int64_t words = 0;
  for (size_t i = 0; i < n; i++) {
    const int64_t m = off[i + 1] - off[i];
    if (off[0] != 0 || m < 0 || m >= (1 << 26)) {
      ctx->err = "anchor offsets must start at 0, ascend, and hold fewer than 2^26 anchors per pair";
      return SDF_ERR_INVALID;
    }
    ws_off[i] = words;
    int c = 6;
    if (!threads_only && m < (1 << 20)) {
      const size_t need = sdf::chain_wave_lds_bytes((int)m);
      for (int q = 5; q >= 0; --q)
        if (need <= caps[q]) c = q;
    }
    cls[c].push_back((int32_t)i);
    if (c == 6 && m > 0) {
      int bits = 0;
      for (unsigned v = (unsigned)m - 1u; v; v >>= 1) ++bits;
      words += 12 * m + 4 * ((int64_t)2 << bits);
    }
  }
  if (cls[5].size() > 512) {  // many large pairs: every one of them in flight instead
    for (int32_t i : cls[5]) {
      const int64_t m = off[i + 1] - off[i];
      int bits = 0;
      for (unsigned v = (unsigned)m - 1u; v; v >>= 1) ++bits;
      ws_off[i] = words;
      words += 12 * m + 4 * ((int64_t)2 << bits);
    }
    cls[6].insert(cls[6].end(), cls[5].begin(), cls[5].end());
    cls[5].clear();
  }
  ws_off[n] = words;
  std::vector<int32_t> which;
  size_t cls_first[7];
  for (int c = 0; c < 7; ++c) {
    std::stable_sort(cls[c].begin(), cls[c].end(), [&](int32_t a, int32_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
    cls_first[c] = which.size();
    which.insert(which.end(), cls[c].begin(), cls[c].end());
  }
  const size_t total = (size_t)off[n];
  if (total && (!anchors || !path)) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  SDF_HIP(ctx->ch_an.reserve(total * sizeof(sdf_anchor) + 16));
  SDF_HIP(ctx->ch_off.reserve((n + 1) * 8));
  SDF_HIP(ctx->ch_wsoff.reserve((n + 1) * 8));
  SDF_HIP(ctx->ch_work.reserve((size_t)words * 4 + 16));
  SDF_HIP(ctx->ch_path.reserve(total * 4 + 16));
  SDF_HIP(ctx->ch_bounds.reserve((total + n) * 8));
  SDF_HIP(ctx->ch_nb.reserve(n * 4));
  if (total) SDF_HIP(hipMemcpyAsync(ctx->ch_an.p, anchors, total * sizeof(sdf_anchor), hipMemcpyHostToDevice, st));
  SDF_HIP(hipMemcpyAsync(ctx->ch_off.p, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  SDF_HIP(hipMemcpyAsync(ctx->ch_wsoff.p, ws_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
  SDF_HIP(ctx->ch_which.reserve(n * 4 + 16));
  SDF_HIP(hipMemcpyAsync(ctx->ch_which.p, which.data(), n * 4, hipMemcpyHostToDevice, st));
  for (int c = 0; c < 6; ++c)
    if (!cls[c].empty())
      hipLaunchKernelGGL(sdf::chain_wave_kernel, dim3((unsigned)cls[c].size()), dim3(64), caps[c], st,
                         (const sdf_anchor *)ctx->ch_an.p, (const int64_t *)ctx->ch_off.p,
                         (const int32_t *)ctx->ch_which.p + cls_first[c], max_chain_gap, match_chain_score,
                         (int32_t *)ctx->ch_path.p, (int32_t *)ctx->ch_bounds.p, (int32_t *)ctx->ch_nb.p);
  if (!cls[6].empty())
    hipLaunchKernelGGL(sdf::chain_kernel, dim3((unsigned)((cls[6].size() + 63) / 64)), dim3(64), 0, st,
                       (const sdf_anchor *)ctx->ch_an.p, (const int64_t *)ctx->ch_off.p, (const int64_t *)ctx->ch_wsoff.p,
                       (int)cls[6].size(), max_chain_gap, match_chain_score, (int32_t *)ctx->ch_work.p, (int32_t *)ctx->ch_path.p,
                       (int32_t *)ctx->ch_bounds.p, (int32_t *)ctx->ch_nb.p, (const int32_t *)ctx->ch_which.p + cls_first[6]);
  SDF_HIP(hipGetLastError());
  if (total) SDF_HIP(hipMemcpyAsync(path, ctx->ch_path.p, total * 4, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipMemcpyAsync(bounds, ctx->ch_bounds.p, (total + n) * 8, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipMemcpyAsync(nbound, ctx->ch_nb.p, n * 4, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
  return SDF_OK;
}

// Test hook: a script of tree operations on chain.hip's device tree (host buffers; one GPU thread).  Returns the number
// of tree nodes (state[i] = node i's p pointer, i < min(nodes, state_cap)) or a negative error code.
extern "C" int sdf_debug_chain_tree_script(sdf_ctx *ctx, const int32_t *pts, int n, const int32_t *ops, int nops, int32_t *out,
                                           int32_t *state, int state_cap) {
  if (!ctx || !pts || n < 1 || nops < 0 || (nops && (!ops || !out))) return SDF_ERR_INVALID;
  ctx->err.clear();
  int bits = 0;
  for (unsigned v = (unsigned)n - 1u; v; v >>= 1) ++bits;
  const int size = (1 << bits) << 1;
  SDF_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t w_pts = (size_t)2 * n, w_ops = (size_t)5 * std::max(nops, 1), w_work = (size_t)4 * n + (size_t)4 * size,
               w_out = (size_t)2 * std::max(nops, 1);
  SDF_HIP(ctx->ch_work.reserve((w_pts + w_ops + w_work + w_out + size) * 4 + 64));
  int32_t *d = (int32_t *)ctx->ch_work.p;
  int32_t *d_pts = d, *d_ops = d_pts + w_pts, *d_work = d_ops + w_ops, *d_out = d_work + w_work, *d_state = d_out + w_out;
  SDF_HIP(hipMemcpyAsync(d_pts, pts, w_pts * 4, hipMemcpyHostToDevice, st));
  if (nops) SDF_HIP(hipMemcpyAsync(d_ops, ops, (size_t)5 * nops * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(sdf::chain_tree_script_kernel, dim3(1), dim3(64), 0, st, d_pts, n, d_ops, nops, d_work, size, d_out, d_state);
  SDF_HIP(hipGetLastError());
  if (nops) SDF_HIP(hipMemcpyAsync(out, d_out, (size_t)2 * nops * 4, hipMemcpyDeviceToHost, st));
  if (state && state_cap > 0)
    SDF_HIP(hipMemcpyAsync(state, d_state, (size_t)std::min(size, state_cap) * 4, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
  return size;
}

// ---- per-alignment columns of `stats generate` (reference: src/stats_main.cc:228-270) -------------------
extern "C" int sdf_stats_columns_device(sdf_ctx *ctx, const sdf_stats_task *d_tasks, size_t n, const char *d_seq_pool,
                                        const uint32_t *d_cigar_pool, sdf_stats_cols *d_out, void *stream) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (n >= ((size_t)1 << 31) || (n && (!d_tasks || !d_out))) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  if (n == 0) return SDF_OK;
  SDF_HIP(hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  // the list for the segments of long alignments (stats_cols.hip): 2^18 segments of 512 runs; an alignment that finds it
  // full is counted by its own wavefront
  const unsigned kItems = ctx->stats_items;
  SDF_HIP(ctx->st_items.reserve((size_t)kItems * sizeof(sdf::StatsItem) + 64));
  unsigned *d_counter = reinterpret_cast<unsigned *>((char *)ctx->st_items.p + (size_t)kItems * sizeof(sdf::StatsItem));
  SDF_HIP(hipMemsetAsync(d_counter, 0, sizeof(unsigned), st));
  const unsigned group_max = ctx->cfg.stats_group_max >= 0 ? (unsigned)ctx->cfg.stats_group_max : sdf::STATS_GROUP_MAX;
  static_assert(sdf::STATS_WAVES == 4, "a workgroup is the four wavefronts of four consecutive alignments");
  hipLaunchKernelGGL(sdf::stats_columns_kernel, dim3((unsigned)((n + sdf::STATS_WAVES - 1) / sdf::STATS_WAVES)),
                     dim3(64 * sdf::STATS_WAVES), 0, st, d_tasks, (int)n, d_seq_pool, d_cigar_pool, d_out,
                     (sdf::StatsItem *)ctx->st_items.p, d_counter, kItems, group_max);
  hipLaunchKernelGGL(sdf::stats_segments_kernel, dim3(2048), dim3(64 * sdf::STATS_WAVES), 0, st,
                     (const sdf::StatsItem *)ctx->st_items.p, d_counter, kItems, d_seq_pool, d_cigar_pool, d_out);
  SDF_HIP(hipGetLastError());
  if (!stream) SDF_HIP(hipStreamSynchronize(st));
  return SDF_OK;
}

extern "C" int sdf_stats_columns_batch(sdf_ctx *ctx, const sdf_stats_task *tasks, size_t n, const char *seq_pool,
                                       size_t pool_bytes, const uint32_t *cigar_pool, size_t cigar_words,
                                       sdf_stats_cols *out) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (n >= ((size_t)1 << 31) || (n && (!tasks || !out)) || (!seq_pool && pool_bytes) || (!cigar_pool && cigar_words)) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  for (size_t i = 0; i < n; i++) {
    const sdf_stats_task &t = tasks[i];
    if (t.a_len > (1u << 24) || t.b_len > (1u << 24)) {
      ctx->err = "stats columns implement sequences up to 16 Mb";
      return SDF_ERR_UNSUPPORTED;
    }
    if (t.a_off > pool_bytes || t.a_len > pool_bytes - t.a_off || t.b_off > pool_bytes || t.b_len > pool_bytes - t.b_off ||
        t.cigar_off > cigar_words || t.n_cigar > cigar_words - t.cigar_off || t.n_cigar >= (1u << 31)) {
      ctx->err = "alignment " + std::to_string(i) + ": sequence or CIGAR range outside its pool";
      return SDF_ERR_INVALID;
    }
  }
  if (n == 0) return SDF_OK;
  SDF_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // (long alignments are cut into segments on the device: stats_cols.hip)
  const sdf_stats_task *up = tasks;
  const size_t nup = n;
  sdf_stats_cols *down = out;
  SDF_HIP(ctx->st_tasks.reserve(nup * sizeof(sdf_stats_task)));
  SDF_HIP(ctx->st_pool.reserve(pool_bytes + 16));
  SDF_HIP(ctx->st_cig.reserve(cigar_words * 4 + 16));
  SDF_HIP(ctx->st_out.reserve(nup * sizeof(sdf_stats_cols)));
  SDF_HIP(hipMemcpyAsync(ctx->st_tasks.p, up, nup * sizeof(sdf_stats_task), hipMemcpyHostToDevice, st));
  if (pool_bytes) SDF_HIP(hipMemcpyAsync(ctx->st_pool.p, seq_pool, pool_bytes, hipMemcpyHostToDevice, st));
  if (cigar_words) SDF_HIP(hipMemcpyAsync(ctx->st_cig.p, cigar_pool, cigar_words * 4, hipMemcpyHostToDevice, st));
  const int rc = sdf_stats_columns_device(ctx, (const sdf_stats_task *)ctx->st_tasks.p, nup, (const char *)ctx->st_pool.p,
                                          (const uint32_t *)ctx->st_cig.p, (sdf_stats_cols *)ctx->st_out.p, st);
  if (rc != SDF_OK) return rc;
  SDF_HIP(hipMemcpyAsync(down, ctx->st_out.p, nup * sizeof(sdf_stats_cols), hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
  static_assert(sizeof(sdf_stats_cols) == 16 * sizeof(int32_t), "sdf_stats_cols is sixteen counters");
  for (size_t i = 0; i < n; i++)
    if (out[i].flags) {
      ctx->err = "alignment " + std::to_string(i) + ": the CIGAR does not fit its sequences";
      return SDF_ERR_INVALID;
    }
  return SDF_OK;
}

// ---- one-task drop-in with the reference's exact signature (extern/ksw2.h:50) -----------------
namespace {
std::mutex g_mu;
sdf_ctx *g_ctx = nullptr;
}  // namespace

extern "C" void sdf_ksw_extz2(void * /*km*/, int qlen, const uint8_t *query, int tlen,
                              const uint8_t *target, int8_t m, const int8_t *mat, int8_t q, int8_t e,
                              int w, int zdrop, int flag, sdf_ksw_extz_t *ez) {
  ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
  ez->max = 0;
  ez->score = ez->mqe = ez->mte = SDF_NEG_INF;
  ez->n_cigar = 0;
  ez->m_cigar = 0;
  ez->zdropped = 0;
  ez->cigar = 0;
  if (m <= 0 || qlen <= 0 || tlen <= 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx) {
    const char *dv = getenv("SDF_DEVICE");
    g_ctx = sdf_create(dv ? atoi(dv) : 0, (size_t)1 << 30);
    if (!g_ctx) {
      fprintf(stderr, "sdf_ksw_extz2: %s\n", sdf_last_error(nullptr));
      exit(120);
    }
  }
  sdf_scoring sc;
  memset(&sc, 0, sizeof(sc));
  sc.m = m;
  if (m == 5) memcpy(sc.mat, mat, 25);
  sc.gapo = q;
  sc.gape = e;
  std::vector<uint8_t> pool((size_t)qlen + tlen);
  memcpy(pool.data(), query, qlen);
  memcpy(pool.data() + qlen, target, tlen);
  sdf_task t;
  memset(&t, 0, sizeof(t));
  t.q_off = 0;
  t.t_off = qlen;
  t.qlen = qlen;
  t.tlen = tlen;
  t.w = w;
  t.zdrop = zdrop;
  t.flag = flag;
  sdf_result r;
  const size_t cap = (size_t)qlen + tlen + 2;
  uint32_t *cig = (uint32_t *)malloc(cap * 4);
  size_t used = 0;
  uint32_t want = SDF_WANT_ALL;
  if (flag & SDF_FLAG_SCORE_ONLY) want &= ~SDF_WANT_CIGAR;
  int rc = sdf_extz2_batch(g_ctx, &sc, &t, 1, pool.data(), pool.size(), want, &r, cig, cap, &used);
  if (rc != SDF_OK) {
    fprintf(stderr, "sdf_ksw_extz2: %s (rc=%d)\n", sdf_last_error(g_ctx), rc);
    exit(120);
  }
  ez->max = (uint32_t)r.max;
  ez->zdropped = (uint32_t)r.zdropped;
  ez->max_q = r.max_q;
  ez->max_t = r.max_t;
  ez->mqe = r.mqe;
  ez->mqe_t = r.mqe_t;
  ez->mte = r.mte;
  ez->mte_q = r.mte_q;
  ez->score = r.score;
  ez->n_cigar = r.n_cigar;
  if (r.n_cigar > 0) {
    ez->cigar = cig;
    ez->m_cigar = (int64_t)cap;
    if (r.cigar_off) memmove(cig, cig + r.cigar_off, (size_t)r.n_cigar * 4);
  } else {
    free(cig);
  }
}

// ---- planner without a device (not part of the public header; tests/test_planner.py) -------------------------------
// Cuts and plans a batch exactly like sdf_extz2_batch_device does (same code, same thread pool) and reports, per input
// task: chunk index (-1: not run), launch class `bs`, nreg, pad_, dir_off, cig_slot, partner (index of the task it
// shares a wavefront with, -1 none), and per chunk {heavy, tasks, launches, dir_bytes, region capacity}.
extern "C" int sdf_debug_plan(const sdf_scoring *sc, const sdf_task *tasks, size_t n, uint32_t want, size_t ws_budget,
                              int max_dyn_lds, int nthreads, int64_t *per_task /* n x 7 */, int64_t *per_chunk /* cap x 5 */,
                              size_t chunk_cap, size_t *nchunks) {
  sdf_ctx tmp;  // only err / flags are used: no HIP call is made here
  tmp.max_dyn_lds = max_dyn_lds;
  PlanEnv env;
  env.tasks = tasks;
  env.n = n;
  env.want = want;
  env.want_cigar = (want & SDF_WANT_CIGAR) != 0;
  sdf_config dcfg;  // (no context here: the environment's settings, like a context made now would get)
  {
    char why[256];
    if (sdf_config_from_env(&dcfg, why, sizeof why) != SDF_OK) return SDF_ERR_INVALID;
  }
  env.cfg = &dcfg;
  ScoreK sk;
  if (int rc = make_scorek(&tmp, sc, sk, env.degenerate)) return rc;
  env.gapo = sc->gapo;
  scoring_gates(sc, env);
  env.max_dyn_lds = max_dyn_lds;
  {  // the strip kernels as a context made now would plan them (batch_part; the lane kernel's records need a device)
    const int qe2 = 2 * (sc->gapo + sc->gape), zm = sc->mat[0] + qe2, zx = sc->mat[1] + qe2;
    env.strip_always = dcfg.strip_always != 0;
    env.strip_cols = (int)dcfg.strip_cols;
    env.chain_min = (size_t)dcfg.chain_min;
    env.strip_ok = dcfg.no_strip == 0 && dcfg.force_general == 0 && !env.degenerate && sc->gapo >= 0 && sc->gape >= 0 && zm >= 0 &&
                   zm <= 127 && zx >= 0 && zx <= 127 && zx - sc->gapo >= 0 && sc->mat[0] >= 0;
  }
  BatchCut cut;
  const char *msg = nullptr;
  const auto tc0 = std::chrono::steady_clock::now();
  std::vector<PlanTask> plan;
  std::vector<int32_t> order;
  // SDF_DEBUG_PLAN_EARLY=1: the cut in two passes with the early start of the heavy chunks (batch_part's, minus the
  // device): buffers by their upper bounds, the heavy chunks planned from the callback
  PlanScratch early_scratch;
  const std::function<int()> early = [&]() -> int {
    plan.resize(std::max<size_t>(n, 1));
    order.resize(std::max<size_t>(cut.order_upper, 2));
    for (size_t ci = 0; ci < cut.chunks.size(); ++ci) {
      plan_chunk(env, cut, cut.chunks[ci], plan.data(), order.data(), early_scratch);
      if (cut.chunks[ci].err) return SDF_ERR_INVALID;
      cut.n_early = ci + 1;
    }
    return SDF_OK;
  };
  {
    WorkerPool cut_pool(std::max(nthreads, 1));
    if (int rc = cut_batch(env, true, ws_budget, cut, &msg, nthreads > 0 ? &cut_pool : nullptr,
                           dcfg.debug_plan_early ? &early : nullptr))
      return rc;
  }
  if (dcfg.debug_plan)
    fprintf(stderr, "[sdf] debug plan: cut %.2f ms (%zu tasks, %d threads, %zu chunks started early)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count(), n, nthreads, cut.n_early);
  const size_t np = std::max<size_t>(cut.ntask_total, 1);
  if (cut.n_early && np > plan.size()) return SDF_ERR_INVALID;  // (the upper bound of the plan records holds)
  if (!cut.n_early) plan.resize(np);
  if (order.size() < std::max<size_t>(cut.order_total, 2)) order.resize(std::max<size_t>(cut.order_total, 2));
  {
    WorkerPool pool(std::max(nthreads, 1));
    ChunkPlanner planner(env, cut, plan.data(), order.data(), &pool, nthreads, cut.n_early);
    for (size_t ci = cut.n_early; ci < cut.chunks.size(); ++ci) planner.wait(ci);
  }
  for (size_t k = 0; k < n * 7; ++k) per_task[k] = -1;
  *nchunks = cut.chunks.size();
  for (size_t ci = 0; ci < cut.chunks.size(); ++ci) {
    const ChunkPlan &c = cut.chunks[ci];
    if (c.err) return SDF_ERR_INVALID;
    if (ci < chunk_cap) {
      int64_t *o = per_chunk + 5 * ci;
      o[0] = c.heavy;
      o[1] = (int64_t)c.cnt;
      o[2] = (int64_t)c.launches.size();
      o[3] = (int64_t)c.dir_bytes;
      o[4] = (int64_t)(c.heavy ? cut.heavy_need : cut.region_need);
    }
    for (const Launch &L : c.launches) {
      const bool pair = (L.bs >= 100 && L.bs < 200) || L.bs == 500;  // (500: strip kernel, two tasks of any geometry)
      const bool chained = L.bs == 604 || L.bs == 608;               // (an entry per block of a pair of tasks: the listed one's partner through zdrop)
      const bool stripes = (L.bs >= 300 && L.bs < 500) || chained;
      for (size_t e = 0; e < L.cnt; ++e) {
        int32_t rel = order[c.ob + L.off + e];
        if (stripes) {  // one entry per stripe: the task is reported at its stripe 0
          if ((uint32_t)rel >> 24) continue;
          rel &= 0xffffff;
        }
        for (int side = 0; side < (chained ? 2 : 1); ++side) {
          const PlanTask &lead = plan[c.pb + rel];
          if (side && lead.zdrop == rel) break;  // (a chain without a partner)
          const PlanTask &p = side ? plan[c.pb + lead.zdrop] : lead;
          int64_t *o = per_task + 7 * (size_t)p.out_idx;
          if (o[0] >= 0 && !(pair && o[6] == p.out_idx)) return SDF_ERR_INVALID;  // listed twice (only a self-pair may be)
          o[0] = (int64_t)ci;
          o[1] = L.bs;
          o[2] = p.nreg;
          o[3] = p.pad_;
          o[4] = p.dir_off;
          o[5] = p.cig_slot;
          o[6] = pair ? plan[c.pb + order[c.ob + L.off + (e ^ 1)]].out_idx : chained ? plan[c.pb + p.zdrop].out_idx : -1;
        }
      }
    }
  }
  return SDF_OK;
}

// ---- debugging aid (not part of the public header): copy the head of the direction workspace ----
extern "C" int sdf_debug_copy_dir(sdf_ctx *ctx, void *host, size_t bytes) {
  if (!ctx || !ctx->dir_ws.p) return SDF_ERR_INVALID;
  if (bytes > ctx->dir_ws.cap) bytes = ctx->dir_ws.cap;
  return hipMemcpy(host, ctx->dir_ws.p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? SDF_OK : SDF_ERR_HIP;
}
