// Host side of the C ABI (include/sedef_hip.h): planning, workspace, launches, timing.
// Replaces the call site of ksw_extz2_sse in align_helper (reference: src/align.cc:39-68) with a
// batched device path.  No CPU fallback exists here: every DP cell is computed by a gfx950 kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "sdf_internal.h"

namespace sdf {
template <int BS, bool GLOBAL, bool PLAIN>
__global__ void extz2_general_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                     sdf_result *, uint8_t *, size_t);
size_t general_lds_bytes(int qlen, int tlen);
template <int NREG, bool STREAM>
__global__ void extz2_wave_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                  sdf_result *);
size_t wave_lds_bytes(int qlen, int tlen, int nreg);
bool wave_fits_whole(int qlen, int tlen, int nreg);
template <int NREG, bool STREAM>
__global__ void extz2_pair_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                  sdf_result *);
size_t pair_lds_bytes(int qlen, int tlen, int nreg);
bool pair_fits_whole(int qlen, int tlen, int nreg);
template <int NREG>
__global__ void extz2_stripe_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                    sdf_result *);
size_t stripe_lds_bytes(int qlen, int nstripe, int nreg);
size_t stripe_dir_bytes(int qlen, int nreg);
template <int LAYOUT>
__global__ void traceback_kernel(const PlanTask *, int, const uint32_t *, const uint8_t *, sdf_result *, uint32_t *);
__global__ void cigar_scan_blocks_kernel(sdf_result *, int, unsigned long long *);
__global__ void cigar_scan_parts_kernel(unsigned long long *, int, unsigned long long *);
__global__ void cigar_scan_add_kernel(sdf_result *, int, const unsigned long long *);
__global__ void cigar_compact_kernel(const PlanTask *, int, const sdf_result *, const uint32_t *,
                                     uint32_t *, unsigned long long);

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
}  // namespace sdf

using namespace sdf;

namespace {

std::string g_err;  // error of the last failed sdf_create

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    static const bool dbg_t = getenv("SDF_DEBUG_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t old = cap;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    // growing means a free (which waits for the device) and an allocation, tens to hundreds of milliseconds for
    // gigabytes: leave half as much again as headroom (at most 8 GiB) so that batches of similar size do not regrow
    size_t want = bytes + std::min<size_t>(bytes / 2, (size_t)8 << 30) + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      want = bytes;
      e = hipMalloc(&p, want);
    }
    if (e == hipSuccess) cap = want;
    if (dbg_t && want >= (64u << 20))
      fprintf(stderr, "[DevBuf %zu -> %zu MiB in %.1f ms]\n", old >> 20, want >> 20,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct HostBuf {  // pinned host memory
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct sdf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  size_t ws_budget = 0;
  hipStream_t dp_stream[2] = {nullptr, nullptr}, tb_stream = nullptr;  // chunk pipeline
  hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};    // more room for launches that end in a tail
  DevBuf dir_ws, stage_ws, plan_buf, order_buf, misc_buf, gstate_buf;
  HostBuf host_plan, host_order;  // pinned staging of the plan
  HostBuf host_pool, host_out;    // pinned staging of the host-buffer entry point (packed sequences; results + CIGARs)
  DevBuf an_pool, an_pairs, an_keys, an_keys2, an_q, an_off, an_flag, an_pos, an_cand, an_out, an_tmp, an_outoff;
  DevBuf ch_an, ch_off, ch_wsoff, ch_work, ch_path, ch_bounds, ch_nb;
  DevBuf h_pool, h_out, h_cig;  // device buffers of the host-buffer entry point
  std::vector<hipEvent_t> events;
  float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0 DP, 1 traceback, 2 compaction, 3 stream total, 4 host planning before the
                                           // first launch, 5 host total, 6 sum of the chunks' DP intervals
  int launches = 0;
  long long paired = 0;  // tasks of the last batch that ran two per wavefront (extz2_pair.hip)
  std::string err;
  int max_dyn_lds = 64 * 1024;
  bool force_general = false;  // SDF_FORCE_GENERAL=1: route everything to the LDS-resident kernel
  bool pipeline = true;        // SDF_PIPELINE=0: one chunk on one stream (isolated kernel timing)
  int stripe_min = 1024;       // SDF_STRIPE_MIN: targets longer than this (and full band) take the stripe kernel
  bool no_stripe = false;      // SDF_NO_STRIPE=1: wide full-band tasks stay on the general kernel (extz2_stripe.hip off)
  bool no_pair = false;        // SDF_NO_PAIR=1: never pack two tasks into one wavefront (extz2_pair.hip)
};

#define SDF_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return SDF_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

extern "C" int sdf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *sdf_last_error(const sdf_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

extern "C" sdf_ctx *sdf_create(int device, size_t workspace_bytes) {
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
  sdf_ctx *ctx = new sdf_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    g_err = "hipStreamCreate failed";
    delete ctx;
    return nullptr;
  }
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  size_t budget = workspace_bytes ? workspace_bytes : (size_t)64 << 30;
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
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<4, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<4, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<8, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_wave_kernel<8, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
#define SDF_PAIR_ATTR(N)                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_pair_kernel<N, false>),  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);         \
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&extz2_pair_kernel<N, true>),   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  SDF_PAIR_ATTR(1) SDF_PAIR_ATTR(2) SDF_PAIR_ATTR(3) SDF_PAIR_ATTR(4) SDF_PAIR_ATTR(6) SDF_PAIR_ATTR(8)
#undef SDF_PAIR_ATTR
  (void)hipGetLastError();
  const char *fg = getenv("SDF_FORCE_GENERAL");
  ctx->force_general = fg && fg[0] == '1';
  const char *np = getenv("SDF_NO_PAIR");
  ctx->no_pair = np && np[0] == '1';
  for (const void *f : {reinterpret_cast<const void *>(&extz2_stripe_kernel<1>),
                        reinterpret_cast<const void *>(&extz2_stripe_kernel<2>),
                        reinterpret_cast<const void *>(&extz2_stripe_kernel<4>)})
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, want_lds);
  (void)hipGetLastError();
  const char *ns = getenv("SDF_NO_STRIPE");
  ctx->no_stripe = ns && ns[0] == '1';
  if (const char *sm = getenv("SDF_STRIPE_MIN")) ctx->stripe_min = std::max(128, atoi(sm));
  const char *pl = getenv("SDF_PIPELINE");
  ctx->pipeline = !(pl && pl[0] == '0');
  if (hipStreamCreateWithFlags(&ctx->dp_stream[0], hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->dp_stream[1], hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->tb_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pipeline = false;
  }
  // (three streams of our own: the runtime multiplexes streams onto GPU_MAX_HW_QUEUES -- default 4 -- hardware
  // queues, and two of ours landing on one queue serialises what the pipeline wants side by side; with the
  // caller's stream that makes four)
  return ctx;
}

extern "C" void sdf_destroy(sdf_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (hipStream_t q : {ctx->stream, ctx->dp_stream[0], ctx->dp_stream[1], ctx->tb_stream, ctx->aux_stream[0],
                        ctx->aux_stream[1], ctx->aux_stream[2], ctx->aux_stream[3]})
    if (q) (void)hipStreamSynchronize(q);
  for (auto ev : ctx->events) (void)hipEventDestroy(ev);
  for (DevBuf *b : {&ctx->an_pool, &ctx->an_pairs, &ctx->an_keys, &ctx->an_keys2, &ctx->an_q, &ctx->an_off, &ctx->an_flag,
                    &ctx->an_pos, &ctx->an_cand, &ctx->an_out, &ctx->an_tmp, &ctx->an_outoff, &ctx->ch_an, &ctx->ch_off,
                    &ctx->ch_wsoff, &ctx->ch_work, &ctx->ch_path, &ctx->ch_bounds, &ctx->ch_nb})
    b->release();
  for (DevBuf *b : {&ctx->dir_ws, &ctx->stage_ws, &ctx->plan_buf, &ctx->order_buf, &ctx->misc_buf, &ctx->gstate_buf,
                    &ctx->h_pool, &ctx->h_out, &ctx->h_cig})
    b->release();
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  for (hipStream_t s : {ctx->dp_stream[0], ctx->dp_stream[1], ctx->tb_stream, ctx->aux_stream[0], ctx->aux_stream[1],
                        ctx->aux_stream[2], ctx->aux_stream[3]})
    if (s) (void)hipStreamDestroy(s);
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
  int min_sc = sc->mat[1];
  for (int t = 1; t < sc->m * sc->m; ++t) min_sc = std::min<int>(min_sc, sc->mat[t]);
  degenerate = -min_sc > 2 * (q + e);  // reference returns before any work (:81)
  return SDF_OK;
}

hipEvent_t next_event(sdf_ctx *ctx, size_t &cursor) {
  if (cursor == ctx->events.size()) {
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    ctx->events.push_back(ev);
  }
  return ctx->events[cursor++];
}

}  // namespace

extern "C" int sdf_extz2_batch_device(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks,
                                      size_t n, const uint32_t *d_pool, uint32_t want,
                                      sdf_result *d_out, uint32_t *d_cig, size_t cigar_cap,
                                      size_t *cigar_used, void *stream_) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  for (float &m : ctx->ms) m = 0.f;
  ctx->launches = 0;
  ctx->paired = 0;
  const auto host_t0 = std::chrono::steady_clock::now();
  if (cigar_used) *cigar_used = 0;
  if (n == 0) return SDF_OK;
  if (!tasks || !d_out || n > 0x7fffffffu) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
  ScoreK sk;
  bool degenerate = false;
  if (int rc = make_scorek(ctx, sc, sk, degenerate)) return rc;
  const bool want_cigar = (want & SDF_WANT_CIGAR) != 0;

  // ---- pre-pass: validation, CIGAR staging size, chunk boundaries ----
  // The batch is cut into chunks that are planned, uploaded and launched one after the other: while the GPU
  // runs chunk i the host plans chunk i+1, the big DP launches of consecutive chunks alternate between two
  // streams (the next chunk fills the CUs while the previous one drains), launches of a few tasks (each a full
  // task latency long) go, longest first, to whichever of the four streams has the least work queued, and the
  // traceback of a chunk runs on the fourth stream next to the following chunk's DP.
  // Direction-flag regions rotate over `nreg_ws` slices of the workspace.
  // (small batches stay on the caller's stream -- unless they hold long tasks: their launch classes, each as long
  // as its longest task, then run side by side on the other streams like those of a large batch)
  bool any_long = false;
  if (n < 2048)
    for (size_t k = 0; k < n && !any_long; ++k) any_long = tasks[k].qlen + (int64_t)tasks[k].tlen >= 3000;
  const bool pipelined = ctx->pipeline && (n >= 2048 || any_long);
  size_t nch = 1;
  if (pipelined && n >= 32768) nch = std::min<size_t>(4, n / 16384);  // a traceback launch is ~2 ms of latency
  const size_t max_regions = nch > 1 ? 4 : 1;
  // Heavy tasks (>= 1 MB of direction flags: long sequences, one workgroup or wavefront busy for milliseconds)
  // leave the chunk rotation: they are planned and launched FIRST, all together, with a workspace slice of
  // their own, and run next to the chunks of ordinary tasks instead of ending each chunk
  // with a long tail.
  const size_t first_target = nch > 1 ? std::max<size_t>(4096, n / (4 * nch + 1)) : n;
  const size_t chunk_target = nch > 1 ? (n - first_target + nch - 1) / nch : n;
  struct Chunk {
    size_t s, e;
    bool heavy;
  };
  std::vector<Chunk> chunks, heavy_chunks;
  std::vector<size_t> bound(n, 0);  // upper bound of each task's direction flags, whichever kernel takes it
  int64_t stage_total = 0;
  size_t n_heavy = 0, heavy_bytes = 0;
  {
    // (four host threads for batches of hundreds of thousands of tasks: this pass is all the planning the GPU
    // waits for besides the first chunk)
    struct Part {
      int64_t stage = 0;
      size_t nh = 0, hb = 0;
      bool bad = false;
    };
    auto scan = [&](size_t lo, size_t hi, Part &pt) {
      for (size_t k = lo; k < hi; ++k) {
        const sdf_task &t = tasks[k];
        if (t.flag & (SDF_FLAG_GENERIC_SC | SDF_FLAG_APPROX_MAX | SDF_FLAG_APPROX_DROP | 0x300)) {
          pt.bad = true;
          return;
        }
        if (t.qlen > 0 && t.tlen > 0 && !degenerate && want_cigar && !(t.flag & SDF_FLAG_SCORE_ONLY)) {
          pt.stage += (int64_t)t.qlen + t.tlen + 2;
          const int w = t.w < 0 ? std::max(t.qlen, t.tlen) : t.w;
          const int ncol16 = ((std::min(std::min(t.qlen, t.tlen), w + 1) + 15) / 16 + 1) * 16;
          const size_t nrow = (size_t)t.qlen + t.tlen - 1;
          const int need = std::min(ncol16 + 32, (t.tlen + 15) / 16 * 16);
          size_t bd = (nrow * (size_t)ncol16 + 16 + 255) & ~(size_t)255;
          if (need <= 1024) bd = std::max(bd, (nrow + 15) / 16 * (size_t)((need + 127) / 128) * 1024);
          bound[k] = bd;
          if (bd >= ((size_t)1 << 20)) {
            ++pt.nh;
            pt.hb += bd;
          }
        }
      }
    };
    const int nthr = n >= 200000 ? 4 : 1;
    Part parts[4];
    std::vector<std::thread> thr;
    for (int q = 1; q < nthr; ++q) thr.emplace_back(scan, n * q / nthr, n * (q + 1) / nthr, std::ref(parts[q]));
    scan(0, n / nthr, parts[0]);
    for (auto &th : thr) th.join();
    for (int q = 0; q < nthr; ++q) {
      if (parts[q].bad) {
        ctx->err = "task flag not implemented on the GPU path (generic scoring / approximate max)";
        return SDF_ERR_UNSUPPORTED;
      }
      stage_total += parts[q].stage;
      n_heavy += parts[q].nh;
      heavy_bytes += parts[q].hb;
    }
  }
  // (a batch that is mostly long tasks is an ordinary batch of long tasks: nothing to take out of the rotation)
  const bool split_heavy = pipelined && n_heavy * 4 <= n;
  // the heavy slice: what the heavy tasks need, up to half of the workspace
  const size_t heavy_budget = split_heavy && n_heavy ? std::min(heavy_bytes + 256, ctx->ws_budget / 2) : 0;
  const size_t region_budget = (ctx->ws_budget - heavy_budget) / max_regions;
  std::vector<uint8_t> heavy(split_heavy ? n : 0, 0);
  size_t region_need = 16, heavy_need = 0;
  {
    size_t s = 0, acc = 0, cnt = 0, hs = 0, hacc = 0;
    bool hany = false;
    for (size_t k = 0; k < n; ++k) {
      const size_t bd = bound[k];
      if (split_heavy && bd >= ((size_t)1 << 20)) {
        heavy[k] = 1;
        if (hany && hacc + bd > heavy_budget) {
          heavy_chunks.push_back({hs, k, true});
          heavy_need = std::max(heavy_need, hacc);
          hs = k;
          hacc = 0;
        }
        if (!hany) hs = k;
        hany = true;
        hacc += bd;
        continue;
      }
      // the first chunk is a quarter of the others: the GPU starts after a quarter of the planning time
      if (k > s && (acc + bd > region_budget || cnt >= (chunks.empty() && nch > 1 ? first_target : chunk_target))) {
        chunks.push_back({s, k, false});
        region_need = std::max(region_need, acc);
        s = k;
        acc = 0;
        cnt = 0;
      }
      acc += bd;
      ++cnt;
    }
    chunks.push_back({s, n, false});
    region_need = std::max(region_need, acc);
    if (hany) {
      heavy_chunks.push_back({hs, n, true});
      heavy_need = std::max(heavy_need, hacc);
    }
  }
  const float dbg_a = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  region_need = (region_need + 255) & ~(size_t)255;
  heavy_need = (heavy_need + 255) & ~(size_t)255;
  const size_t nreg_ws = std::min(max_regions, chunks.size());
  chunks.insert(chunks.begin(), heavy_chunks.begin(), heavy_chunks.end());  // heavy first
  if (ctx->dir_ws.reserve(region_need * nreg_ws + heavy_need) != hipSuccess) {
    ctx->err = "cannot allocate the direction-matrix workspace";
    (void)hipGetLastError();
    return SDF_ERR_NOMEM;
  }
  SDF_HIP(ctx->stage_ws.reserve((size_t)std::max<int64_t>(stage_total, 4) * 4));
  SDF_HIP(ctx->plan_buf.reserve(n * sizeof(PlanTask)));
  SDF_HIP(ctx->order_buf.reserve(2 * n * sizeof(int32_t)));  // a task paired with itself is listed twice
  SDF_HIP(ctx->misc_buf.reserve(256 + ((n + 1023) / 1024 + 1) * 8));
  SDF_HIP(ctx->host_plan.reserve(n * sizeof(PlanTask)));
  SDF_HIP(ctx->host_order.reserve(2 * n * sizeof(int32_t)));
  PlanTask *const plan = (PlanTask *)ctx->host_plan.p;  // pinned: the uploads below are asynchronous
  int32_t *const order = (int32_t *)ctx->host_order.p;
  PlanTask *d_plan = (PlanTask *)ctx->plan_buf.p;
  int32_t *d_order = (int32_t *)ctx->order_buf.p;
  unsigned long long *d_total = (unsigned long long *)ctx->misc_buf.p;
  uint8_t *d_dir = (uint8_t *)ctx->dir_ws.p;
  uint32_t *d_stage = (uint32_t *)ctx->stage_ws.p;

  const float dbg_b = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  size_t evc = 0;
  hipEvent_t ev_begin = next_event(ctx, evc);
  hipLaunchKernelGGL(reset_results_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_out,
                     (int)n);
  SDF_HIP(hipEventRecord(ev_begin, st));
  if (pipelined)
    for (hipStream_t s : {ctx->dp_stream[0], ctx->dp_stream[1], ctx->tb_stream}) SDF_HIP(hipStreamWaitEvent(s, ev_begin, 0));

  struct Cls {
    int bs;  // 64 / 256: general kernel with that many threads; 1, 2, 4, 8: wave kernel with NREG; 100 + NREG: pair
             // kernel; 1000: general kernel with its state in HBM
    size_t lds;       // class key
    size_t need_max;  // largest real requirement in the class: what the launch asks for
    std::vector<int32_t> idx;
    double est = 0;   // duration estimate of the launch: its longest task (cells / per-workgroup rate of the kernel)
    int kmax = 0;     // stripe kernel: wavefronts per workgroup (largest stripe count in the class)
  };
  struct ChunkEv {
    hipEvent_t dp0, dpe[8], tb0, tb1;  // plan uploaded; end of the DP launches per stream; traceback (begin, end)
  };
  std::vector<ChunkEv> cev(chunks.size());
  std::vector<int32_t> win_need, partner;
  std::vector<std::pair<int32_t, int32_t>> table;
  std::vector<Cls> cls;
  size_t np = 0;           // planned tasks so far (= index of the next PlanTask)
  size_t nord = 0;         // launch-order entries so far
  int64_t stage_words = 0;
  float plan_first_ms = 0.f;
  double qload[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // estimated DP work queued on each stream during this call

  std::vector<size_t> normal_ids;  // chunk indices of the ordinary chunks, in launch order
  const bool have_heavy = !chunks.empty() && chunks[0].heavy;
  for (size_t ci = 0; ci < chunks.size(); ++ci) {
    const size_t pb = np;    // first PlanTask of this chunk
    const size_t ob = nord;  // first launch-order entry of this chunk
    const bool heavy_chunk = chunks[ci].heavy;
    // ---- plan the chunk ----
    win_need.clear();
    int snreg = 0;  // widest stripe any stripe task of the chunk needs
    for (size_t k = chunks[ci].s; k < chunks[ci].e; ++k) {
      const sdf_task &t = tasks[k];
      if (split_heavy && (heavy[k] != 0) != heavy_chunk) continue;
      if (t.qlen <= 0 || t.tlen <= 0 || degenerate) continue;  // reference early return (:57,:81)
      PlanTask p;
      p.q_word = t.q_off;
      p.t_word = t.t_off;
      p.qlen = t.qlen;
      p.tlen = t.tlen;
      p.w = t.w < 0 ? std::max(t.qlen, t.tlen) : t.w;
      p.zdrop = t.zdrop;
      p.flag = t.flag | (want_cigar ? 0 : SDF_FLAG_SCORE_ONLY);
      int nc = std::min(t.qlen, t.tlen);
      nc = (std::min(nc, p.w + 1) + 15) / 16 + 1;
      p.ncol16 = nc * 16;
      p.out_idx = (int32_t)k;
      p.pad_ = 0;
      // register-resident wave kernel when only CIGAR/score/mte are wanted and the shape fits
      p.nreg = 0;
      int wneed = 0;
      bool plain_ok = false;
      {
        const int nrow = t.qlen + t.tlen - 1;
        Band bl;
        const bool band_whole = (p.w >= 1 || nrow == 1) && band_of(nrow - 1, t.qlen, t.tlen, p.w, bl);
        const bool plain = !(want & SDF_WANT_EXT) && t.zdrop < 0 &&
                           !(t.flag & (SDF_FLAG_RIGHT | SDF_FLAG_EXTZ_ONLY)) && sc->gapo >= 0 && band_whole;
        plain_ok = plain && !ctx->force_general;
        if (plain && !ctx->force_general) {
          // window slots: one 16-row block of slack below, the score refresh overshoot above -- but never
          // beyond the target's last 16-cell block (cells past it are not part of any window)
          const int need = std::min(p.ncol16 + 32, (t.tlen + 15) / 16 * 16);
          const int nreg = need <= 128 ? 1 : need <= 256 ? 2 : need <= 512 ? 4 : need <= 1024 ? 8 : 0;
          if (nreg && wave_lds_bytes(t.qlen, t.tlen, nreg) <= (size_t)ctx->max_dyn_lds) {
            p.nreg = nreg;
            wneed = need;
          }
        }
      }
      win_need.push_back(wneed);
      p.dir_off = 0;
      p.cig_cap = (p.flag & SDF_FLAG_SCORE_ONLY) ? 0 : t.qlen + t.tlen + 2;
      p.cig_slot = stage_words;
      stage_words += p.cig_cap;
      if ((!p.nreg || ctx->stripe_min < 1024) && plain_ok && !ctx->no_stripe && p.w >= std::max(t.qlen, t.tlen) &&
          t.tlen > ctx->stripe_min && t.tlen <= 8192) {
        // wide full-band task: a workgroup of wavefronts, one per stripe of 128 * nreg target positions
        const int nreg = t.tlen <= 2048 ? 1 : t.tlen <= 4096 ? 2 : 4;
        const int nst = (t.tlen + 128 * nreg - 1) / (128 * nreg);
        if (stripe_lds_bytes(t.qlen, nst, nreg) <= (size_t)ctx->max_dyn_lds) {
          p.nreg = nreg;
          p.pad_ = 5;
          snreg = std::max(snreg, nreg);
        }
      }
      if (!p.nreg) {
        // general kernel: state in LDS, or in an HBM scratch slab when it does not fit; the PLAIN flavour (packed
        // recurrence, H along the band edge only) when nothing but CIGAR / score / mte is wanted and the window
        // is wide enough for the 256- or 1024-thread instantiation
        const bool hbm = general_lds_bytes(t.qlen, t.tlen) > (size_t)ctx->max_dyn_lds;
        const int width = std::min(p.ncol16, (t.tlen + 15) / 16 * 16);
        if (plain_ok && !hbm && width > 256) p.pad_ = 3;
        else if (plain_ok && hbm && width > 1024) p.pad_ = 4;
        else p.pad_ = hbm ? 1 : 0;
      }
      plan[np++] = p;
    }
    const size_t cnt = np - pb;
    if (snreg) {  // the stripe tasks of a chunk share one stripe width (the widest any of them needs): one launch,
                  // all of them side by side, instead of one launch per width queued behind each other
      for (size_t k = pb; k < np; ++k) {
        PlanTask &p = plan[k];
        if (p.pad_ != 5) continue;
        p.nreg = snreg;
        // a last stripe of one cell would need the H of the cell under the target's end from its neighbour:
        // such a task stays on the general kernel
        if (p.tlen % (128 * snreg) == 1) {
          p.nreg = 0;
          const bool hbm = general_lds_bytes(p.qlen, p.tlen) > (size_t)ctx->max_dyn_lds;
          p.pad_ = hbm ? 4 : 3;
        }
      }
    }
    if (cnt == 0) {
      cev[ci] = ChunkEv{};
      continue;
    }
    PlanTask *cp = plan + pb;  // chunk-relative indexing below

    // Pair kernel: two wave-eligible tasks of the chunk with the same (qlen, tlen, w, flag) and a window of at
    // most 512 slots share a wavefront (extz2_pair.hip).  partner[k] = the other task, or -1.  One pass with an
    // open-addressing table keyed by the geometry: entry = (first task seen with the key, the task of that key
    // still waiting for a partner or -1).
    partner.assign(cnt, -1);
    if (!ctx->no_pair && !ctx->force_general) {
      size_t cap = 64;
      while (cap < 2 * cnt) cap *= 2;
      table.assign(cap, {-1, -1});
      for (size_t k = 0; k < cnt; ++k) {
        PlanTask &y = cp[k];
        if (!y.nreg || y.pad_ != 0 || win_need[k] > 512) continue;  // wave-kernel tasks only
        uint64_t h = ((uint64_t)(uint32_t)y.qlen * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)y.tlen * 0xC2B2AE3D27D4EB4Full) ^
                     ((uint64_t)(uint32_t)y.w * 0x165667B19E3779F9ull) ^ ((uint64_t)(uint32_t)y.flag << 40);
        h ^= h >> 29;
        for (size_t slot = (size_t)h & (cap - 1);; slot = (slot + 1) & (cap - 1)) {
          auto &e = table[slot];
          if (e.first < 0) {
            e = {(int32_t)k, (int32_t)k};
            break;
          }
          const PlanTask &x = cp[e.first];
          if (x.qlen != y.qlen || x.tlen != y.tlen || x.w != y.w || x.flag != y.flag) continue;
          if (e.second < 0) {
            e.second = (int32_t)k;
            break;
          }
          PlanTask &z = cp[e.second];
          const int regs = (win_need[k] + 63) / 64;
          const int nreg = regs <= 4 ? regs : regs <= 6 ? 6 : 8;
          if (pair_lds_bytes(y.qlen, y.tlen, nreg) > (size_t)ctx->max_dyn_lds) break;
          z.nreg = y.nreg = nreg;
          z.pad_ = y.pad_ = 2;
          partner[k] = e.second;
          partner[e.second] = (int32_t)k;
          ctx->paired += 2;
          e.second = -1;
          break;
        }
      }
      // a task left without a partner is paired with itself (both halves compute the same task and write the
      // same bytes) instead of occupying a launch of its own for a whole task latency
      for (auto &e : table) {
        if (e.second < 0) continue;
        PlanTask &y = cp[e.second];
        const int regs = (win_need[e.second] + 63) / 64;
        const int nreg = regs <= 4 ? regs : regs <= 6 ? 6 : 8;
        if (pair_lds_bytes(y.qlen, y.tlen, nreg) > (size_t)ctx->max_dyn_lds) continue;
        y.nreg = nreg;
        y.pad_ = 2;
        partner[e.second] = e.second;
      }
    }
    // launch classes: (kernel, LDS bytes rounded to a power of two)
    struct Launch {
      int bs;
      size_t lds;
      size_t off, cnt;
      double est;
      int kmax;
    };
    std::vector<Launch> launches;
    {
      cls.clear();
      size_t dir_acc = 0;  // direction-flag layout inside this chunk's workspace region, in the same pass
      for (size_t k = 0; k < cnt; ++k) {
        PlanTask &p = cp[k];
        {
          size_t need = 0;
          if (!(p.flag & SDF_FLAG_SCORE_ONLY)) {
            const size_t nblk = (size_t)((p.qlen + p.tlen - 1 + 15) / 16);
            if (p.pad_ == 2) need = nblk * (size_t)p.nreg * 512;
            else if (p.pad_ == 5)
              need = (size_t)((p.tlen + 128 * p.nreg - 1) / (128 * p.nreg)) * stripe_dir_bytes(p.qlen, p.nreg);
            else if (p.nreg) need = nblk * (size_t)p.nreg * 1024;
            else need = ((size_t)(p.qlen + p.tlen - 1) * (size_t)p.ncol16 + 16 + 255) & ~(size_t)255;
          }
          p.dir_off = (int64_t)dir_acc;
          dir_acc += need;
        }
        const int width = std::min(p.ncol16, (p.tlen + 15) / 16 * 16);
        int bs = width > 1024 ? 1024 : width > 256 ? 256 : 64;  // 4 cells per thread and pass over the row
        size_t lds = 2048, need;
        if (p.pad_ == 2) {
          if (partner[k] < (int32_t)k) continue;  // placed together with its partner
          // 100 + NREG; + 10 for the streamed-window instantiation (sequences longer than the LDS windows)
          bs = 100 + p.nreg + (pair_fits_whole(p.qlen, p.tlen, p.nreg) ? 0 : 10);
          need = pair_lds_bytes(p.qlen, p.tlen, p.nreg);
          lds = 8192;
          while (lds < need) lds *= 2;
        } else if (p.pad_ == 5) {
          bs = 200 + p.nreg;  // stripe kernel; LDS by the stripe count and the query length
          need = stripe_lds_bytes(p.qlen, (p.tlen + 128 * p.nreg - 1) / (128 * p.nreg), p.nreg);
          lds = 32768;
          while (lds < need) lds *= 2;
        } else if (p.nreg) {
          // NREG; + 10 for the streamed-window instantiation (sequences longer than the LDS windows)
          bs = p.nreg + (wave_fits_whole(p.qlen, p.tlen, p.nreg) ? 0 : 10);
          // one class for everything up to 6 KiB (>= 6 waves/SIMD either way), powers of two above
          need = wave_lds_bytes(p.qlen, p.tlen, p.nreg);
          lds = 6144;
          while (lds < need) lds *= 2;
        } else if (p.pad_ == 1 || p.pad_ == 4) {  // HBM-resident state: one class, slab = largest requirement
          bs = p.pad_ == 4 ? 2001 : width > 1024 ? 1001 : 1000;
          need = general_lds_bytes(p.qlen, p.tlen);
          lds = (size_t)1 << 40;
        } else {
          if (p.pad_ == 3) bs += 2000;  // PLAIN flavour: 2256 / 3024
          need = general_lds_bytes(p.qlen, p.tlen);
          while (lds < need) lds *= 2;
        }
        const bool hbm_cls = bs == 1000 || bs == 1001 || bs == 2001;
        if (!hbm_cls && lds > (size_t)ctx->max_dyn_lds) lds = ctx->max_dyn_lds;
        Cls *c = nullptr;
        for (auto &x : cls)
          if (x.bs == bs && x.lds == lds) c = &x;
        if (!c) {
          cls.push_back({bs, lds, 0, {}});
          c = &cls.back();
        }
        c->need_max = std::max(c->need_max, need);
        if (p.pad_ == 5) c->kmax = std::max(c->kmax, (p.tlen + 128 * p.nreg - 1) / (128 * p.nreg));
        {  // rough per-workgroup rates: general 64 / 256 / 1024 threads, HBM state, wave, pair
          const double rate = bs == 64 ? 0.03 : bs == 256 ? 0.1 : bs == 1024 ? 0.6 : bs == 2256 ? 0.3 : bs == 3024 ? 0.85
                              : bs == 2001 ? 0.3 : bs >= 1000 ? 0.08 : bs >= 200 ? 2.2 : bs >= 100 ? 0.25 : 0.13;  // (pair classes are 100..118)
          c->est = std::max(c->est, (double)(p.qlen + p.tlen) * (double)p.ncol16 / rate);
        }
        c->idx.push_back((int32_t)k);
        if (p.pad_ == 2) c->idx.push_back(partner[k]);
      }
      // Small classes of one kernel (< 2048 tasks: occupancy is not what limits them, their longest task is) are
      // merged into one launch with the largest LDS size among them: fewer launches queued one behind the other.
      for (size_t a = 0; a < cls.size(); ++a) {
        if (cls[a].idx.empty() || cls[a].idx.size() >= 2048) continue;
        for (size_t b = a + 1; b < cls.size(); ++b) {
          if (cls[b].bs != cls[a].bs || cls[b].idx.empty() || cls[b].idx.size() >= 2048) continue;
          cls[a].lds = std::max(cls[a].lds, cls[b].lds);
          cls[a].need_max = std::max(cls[a].need_max, cls[b].need_max);
          cls[a].est = std::max(cls[a].est, cls[b].est);
          cls[a].kmax = std::max(cls[a].kmax, cls[b].kmax);
          cls[a].idx.insert(cls[a].idx.end(), cls[b].idx.begin(), cls[b].idx.end());
          cls[b].idx.clear();
        }
      }
      if (dir_acc > (heavy_chunk ? heavy_need : region_need)) {
        ctx->err = "internal: direction-flag region overflow";
        return SDF_ERR_INVALID;
      }
      cls.erase(std::remove_if(cls.begin(), cls.end(), [](const Cls &c) { return c.idx.empty(); }), cls.end());
      // inside a launch of few tasks the longest go first too (workgroups are dispatched in order: a long task that
      // starts last is the tail of the launch); pair-kernel entries move as (task, partner) units
      for (auto &c : cls) {
        if (c.idx.size() >= 8192 || c.idx.size() < 3) continue;
        auto work = [&](int32_t k) { return (int64_t)(cp[k].qlen + cp[k].tlen) * cp[k].ncol16; };
        int64_t wmin = work(c.idx[0]), wmax = wmin;
        for (int32_t k : c.idx) {
          const int64_t wk = work(k);
          wmin = std::min(wmin, wk);
          wmax = std::max(wmax, wk);
        }
        if (wmax < 2 * wmin) continue;  // tasks of one size: the order does not matter
        if (c.bs >= 100 && c.bs < 200) {
          std::vector<std::pair<int32_t, int32_t>> pr(c.idx.size() / 2);
          for (size_t q = 0; q < pr.size(); ++q) pr[q] = {c.idx[2 * q], c.idx[2 * q + 1]};
          std::stable_sort(pr.begin(), pr.end(), [&](const std::pair<int32_t, int32_t> &x, const std::pair<int32_t, int32_t> &y) {
            return work(x.first) > work(y.first);
          });
          for (size_t q = 0; q < pr.size(); ++q) {
            c.idx[2 * q] = pr[q].first;
            c.idx[2 * q + 1] = pr[q].second;
          }
        } else {
          std::stable_sort(c.idx.begin(), c.idx.end(), [&](int32_t x, int32_t y) { return work(x) > work(y); });
        }
      }
      // longest launches first so the long tasks start early
      std::sort(cls.begin(), cls.end(), [](const Cls &a, const Cls &b) { return a.est > b.est; });
      size_t cursor = 0;
      for (auto &c : cls) {
        launches.push_back({c.bs,
                            (c.bs == 1000 || c.bs == 1001 || c.bs == 2001) ? ((c.need_max + 255) & ~(size_t)255)
                                         : std::min(c.lds, (c.need_max + 511) & ~(size_t)511),
                            cursor, c.idx.size(), c.est, c.kmax});
        std::copy(c.idx.begin(), c.idx.end(), order + ob + cursor);
        cursor += c.idx.size();
      }
      nord += cursor;
    }
    if (ci == 0)
      plan_first_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();

    // ---- upload and launch the chunk ----
    // Streams: Q[0] the caller's, Q[1], Q[2] the two DP streams, Q[3] traceback.  The big launches (>= 2048 tasks)
    // of ordinary chunks alternate between the DP streams; every other launch (a class of a few tasks ends in a
    // tail as long as its longest task) goes, longest first, to the stream with the least estimated work queued.
    // A heavy chunk uploads and traces back on the caller's stream and uses the workspace slice behind the regions.
    const bool piped = pipelined && !heavy_chunk;
    const size_t nj = normal_ids.size();  // ordinal among the ordinary chunks
    if (pipelined) {
      // extra streams, created the first time they are wanted (a stream is a hardware queue: ~7 ms to set up): one
      // for the alternating tracebacks of a multi-chunk batch, one per launch beyond four for a one-chunk batch
      const size_t want_aux = chunks.size() == 1 ? (launches.size() > 4 ? std::min<size_t>(launches.size() - 4, 4) : 0) : 1;
      for (size_t a = 0; a < want_aux; ++a)
        if (!ctx->aux_stream[a] && hipStreamCreateWithFlags(&ctx->aux_stream[a], hipStreamNonBlocking) != hipSuccess) {
          (void)hipGetLastError();
          ctx->aux_stream[a] = nullptr;
        }
    }
    hipStream_t Q[8] = {st, pipelined ? ctx->dp_stream[0] : st, pipelined ? ctx->dp_stream[1] : st,
                        pipelined ? ctx->tb_stream : st, ctx->aux_stream[0], ctx->aux_stream[1], ctx->aux_stream[2],
                        ctx->aux_stream[3]};
    // Q[4..7]: only the least-loaded-stream assignment below uses them (one-chunk batches without heavy tasks)
    const int ui = piped ? (have_heavy ? 2 : 1 + (int)(nj & 1)) : 0;  // upload stream (and the big launches')
    // (the tracebacks of consecutive ordinary chunks alternate between two streams: the last one starts when its DP
    // ends, not when the previous chunk's walk does)
    hipStream_t stb = piped ? ((nj & 1) && ctx->aux_stream[0] ? ctx->aux_stream[0] : Q[3]) : st;
    uint8_t *dir_reg = heavy_chunk ? d_dir + nreg_ws * region_need : d_dir + (nj % nreg_ws) * region_need;
    hipEvent_t region_ev = nullptr;  // the region's previous user has been traced back
    if (piped && nj >= nreg_ws) region_ev = cev[normal_ids[nj - nreg_ws]].tb1;
    if (!heavy_chunk) normal_ids.push_back(ci);
    SDF_HIP(hipMemcpyAsync(d_plan + pb, cp, cnt * sizeof(PlanTask), hipMemcpyHostToDevice, Q[ui]));
    SDF_HIP(hipMemcpyAsync(d_order + ob, order + ob, (nord - ob) * sizeof(int32_t), hipMemcpyHostToDevice, Q[ui]));
    ChunkEv &ev = cev[ci];
    ev.dp0 = next_event(ctx, evc);
    for (auto &e : ev.dpe) e = nullptr;
    ev.tb0 = next_event(ctx, evc);
    ev.tb1 = next_event(ctx, evc);
    SDF_HIP(hipEventRecord(ev.dp0, Q[ui]));
    bool used[8] = {false, false, false, false, false, false, false, false};
    size_t gs_off = 0;
    {  // HBM state slabs of the very long tasks of this chunk: one allocation, a slice per launch
      size_t gs_total = 0;
      for (const Launch &L : launches)
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
    for (const Launch &L : launches) {
      // with heavy tasks in the batch the streams are divided: Q[0], Q[1] for the heavy launches (tens of
      // milliseconds each), Q[2] (DP) and Q[3] (small launches, traceback) for the ordinary chunks, which would
      // otherwise queue behind them
      int qi = 0;
      if (pipelined) {
        if (piped && L.cnt >= 2048) {
          qi = ui;
        } else if (have_heavy) {
          qi = heavy_chunk ? (qload[1] < qload[0] ? 1 : 0) : (qload[3] < qload[2] ? 3 : 2);
        } else {
          // (a batch of one chunk has nothing else to overlap with: its launches spread over the extra streams too)
          for (int q = 1; q < (chunks.size() == 1 ? 8 : 4); ++q)
            if (Q[q] && qload[q] < qload[qi]) qi = q;
        }
      }
      qload[qi] += L.est;
      hipStream_t sdp = Q[qi];
      if (!used[qi]) {
        used[qi] = true;
        if (pipelined && qi != ui) SDF_HIP(hipStreamWaitEvent(sdp, ev.dp0, 0));  // plan uploaded
        if (region_ev) SDF_HIP(hipStreamWaitEvent(sdp, region_ev, 0));
      }
      const PlanTask *lp = d_plan + pb;
      const int32_t *lo = d_order + ob + L.off;
      if (L.bs == 1)
        hipLaunchKernelGGL((extz2_wave_kernel<1, false>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 11)
        hipLaunchKernelGGL((extz2_wave_kernel<1, true>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 2)
        hipLaunchKernelGGL((extz2_wave_kernel<2, false>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 12)
        hipLaunchKernelGGL((extz2_wave_kernel<2, true>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 4)
        hipLaunchKernelGGL((extz2_wave_kernel<4, false>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 14)
        hipLaunchKernelGGL((extz2_wave_kernel<4, true>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 8)
        hipLaunchKernelGGL((extz2_wave_kernel<8, false>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
      else if (L.bs == 18)
        hipLaunchKernelGGL((extz2_wave_kernel<8, true>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo, d_pool, sk,
                           dir_reg, d_out);
#define SDF_STRIPE_LAUNCH(N)                                                                                  \
  else if (L.bs == 200 + N) hipLaunchKernelGGL(extz2_stripe_kernel<N>, dim3((unsigned)L.cnt), dim3(64 * L.kmax), \
                                               L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out);
      SDF_STRIPE_LAUNCH(1)
      SDF_STRIPE_LAUNCH(2)
      SDF_STRIPE_LAUNCH(4)
#undef SDF_STRIPE_LAUNCH
#define SDF_PAIR_LAUNCH(N)                                                                                     \
  else if (L.bs == 100 + N) hipLaunchKernelGGL((extz2_pair_kernel<N, false>), dim3((unsigned)(L.cnt / 2)), dim3(64), \
                                               L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out);                 \
  else if (L.bs == 110 + N) hipLaunchKernelGGL((extz2_pair_kernel<N, true>), dim3((unsigned)(L.cnt / 2)), dim3(64), \
                                               L.lds, sdp, lp, lo, d_pool, sk, dir_reg, d_out);
      SDF_PAIR_LAUNCH(1)
      SDF_PAIR_LAUNCH(2)
      SDF_PAIR_LAUNCH(3)
      SDF_PAIR_LAUNCH(4)
      SDF_PAIR_LAUNCH(6)
      SDF_PAIR_LAUNCH(8)
#undef SDF_PAIR_LAUNCH
      else if (L.bs == 64)
        hipLaunchKernelGGL((extz2_general_kernel<64, false, false>), dim3((unsigned)L.cnt), dim3(64), L.lds, sdp, lp, lo,
                           d_pool, sk, dir_reg, d_out, (uint8_t *)nullptr, (size_t)0);
      else if (L.bs == 256)
        hipLaunchKernelGGL((extz2_general_kernel<256, false, false>), dim3((unsigned)L.cnt), dim3(256), L.lds, sdp, lp, lo,
                           d_pool, sk, dir_reg, d_out, (uint8_t *)nullptr, (size_t)0);
      else if (L.bs == 1024)
        hipLaunchKernelGGL((extz2_general_kernel<1024, false, false>), dim3((unsigned)L.cnt), dim3(1024), L.lds, sdp, lp, lo,
                           d_pool, sk, dir_reg, d_out, (uint8_t *)nullptr, (size_t)0);
      else if (L.bs == 2256)
        hipLaunchKernelGGL((extz2_general_kernel<256, false, true>), dim3((unsigned)L.cnt), dim3(256), L.lds, sdp, lp, lo,
                           d_pool, sk, dir_reg, d_out, (uint8_t *)nullptr, (size_t)0);
      else if (L.bs == 3024)
        hipLaunchKernelGGL((extz2_general_kernel<1024, false, true>), dim3((unsigned)L.cnt), dim3(1024), L.lds, sdp, lp, lo,
                           d_pool, sk, dir_reg, d_out, (uint8_t *)nullptr, (size_t)0);
      else {  // L.lds = per-workgroup slab bytes in HBM; the chunk's slabs were reserved above
        uint8_t *slabs = (uint8_t *)ctx->gstate_buf.p + gs_off;
        gs_off += L.lds * L.cnt;
        if (L.bs == 2001)
          hipLaunchKernelGGL((extz2_general_kernel<1024, true, true>), dim3((unsigned)L.cnt), dim3(1024), 512, sdp, lp, lo,
                             d_pool, sk, dir_reg, d_out, slabs, L.lds);
        else if (L.bs == 1001)
          hipLaunchKernelGGL((extz2_general_kernel<1024, true, false>), dim3((unsigned)L.cnt), dim3(1024), 512, sdp, lp, lo,
                             d_pool, sk, dir_reg, d_out, slabs, L.lds);
        else
          hipLaunchKernelGGL((extz2_general_kernel<256, true, false>), dim3((unsigned)L.cnt), dim3(256), 512, sdp, lp, lo,
                             d_pool, sk, dir_reg, d_out, slabs, L.lds);
      }
      ++ctx->launches;
    }
    for (int q = 0; q < 8; ++q) {  // the traceback stream collects every stream the chunk's DP ran on
      if (!used[q]) continue;
      ev.dpe[q] = next_event(ctx, evc);
      SDF_HIP(hipEventRecord(ev.dpe[q], Q[q]));
      if (pipelined && Q[q] != stb) SDF_HIP(hipStreamWaitEvent(stb, ev.dpe[q], 0));
    }
    SDF_HIP(hipEventRecord(ev.tb0, stb));
    if (want_cigar) {
      unsigned layouts = 0;  // direction-flag layouts present in the chunk: one traceback instantiation each
      for (size_t k = 0; k < cnt; ++k)
        layouts |= 1u << (cp[k].nreg == 0 ? 0 : cp[k].pad_ == 2 ? 2 : cp[k].pad_ == 5 ? 3 : 1);
      // few tasks: a wavefront per walk (see traceback_kernel)
      // (one walk per wavefront costs 64 times the instruction issue of 64 walks per wavefront: only where the
      // walks have the GPU to themselves, or are few)
      const bool tb_solo = cnt <= (chunks.size() == 1 ? (size_t)8192 : (size_t)1024);
      const dim3 tbg(tb_solo ? (unsigned)cnt : (unsigned)((cnt + 63) / 64));
      const int tbn = tb_solo ? -(int)cnt : (int)cnt;
      // A walk is one step per anti-diagonal, ~1 us each: a launch lasts as long as its longest task.  When a chunk
      // of few tasks mixes layouts, the instantiations run side by side on different streams (each after the
      // chunk's DP, collected again by the chunk's traceback stream) rather than one after the other.
      const bool side_by_side = pipelined && cnt < 32768 && (layouts & (layouts - 1)) != 0;
      int used_tb = 0;
      hipStream_t tbs[3] = {stb, stb, stb};  // (a fourth layout shares the last stream)
      if (side_by_side) {  // (a batch with heavy tasks keeps its stream division: Q[0], Q[1] heavy, Q[2], Q[3] ordinary)
        int j = 1;
        for (int q = 0; q < 4 && j < 3; ++q) {
          if (Q[q] == stb) continue;
          if (have_heavy && (heavy_chunk ? q >= 2 : q < 2)) continue;
          tbs[j++] = Q[q];
        }
      }
      auto tb_on = [&]() -> hipStream_t {
        hipStream_t s2 = tbs[used_tb < 3 ? used_tb : 2];
        if (side_by_side && s2 != stb) (void)hipStreamWaitEvent(s2, ev.tb0, 0);
        ++used_tb;
        return s2;
      };
      if (layouts & 8u)
        hipLaunchKernelGGL(traceback_kernel<3>, tbg, dim3(64), 0, tb_on(), d_plan + pb, tbn, d_pool, dir_reg, d_out, d_stage);
      if (layouts & 4u)
        hipLaunchKernelGGL(traceback_kernel<2>, tbg, dim3(64), 0, tb_on(), d_plan + pb, tbn, d_pool, dir_reg, d_out, d_stage);
      if (layouts & 2u)
        hipLaunchKernelGGL(traceback_kernel<1>, tbg, dim3(64), 0, tb_on(), d_plan + pb, tbn, d_pool, dir_reg, d_out, d_stage);
      if (layouts & 1u)
        hipLaunchKernelGGL(traceback_kernel<0>, tbg, dim3(64), 0, tb_on(), d_plan + pb, tbn, d_pool, dir_reg, d_out, d_stage);
      if (side_by_side)
        for (int j = 1; j < used_tb && j < 3; ++j) {
          if (tbs[j] == stb) continue;
          hipEvent_t e = next_event(ctx, evc);
          SDF_HIP(hipEventRecord(e, tbs[j]));
          SDF_HIP(hipStreamWaitEvent(stb, e, 0));
        }
    }
    SDF_HIP(hipEventRecord(ev.tb1, stb));
  }
  ctx->ms[4] = plan_first_ms;
  if (plan_first_ms > 50.f && getenv("SDF_DEBUG_TIMING"))
    fprintf(stderr, "[slow plan: n=%zu pre-pass %.1f ms, buffers %.1f ms, first chunk planned %.1f ms; chunks %zu heavy %zu]\n", n,
            dbg_a, dbg_b, plan_first_ms, chunks.size(), n_heavy);
  if (pipelined)
    for (auto &ev : cev)
      if (ev.tb1) SDF_HIP(hipStreamWaitEvent(st, ev.tb1, 0));

  hipEvent_t ev_c0 = next_event(ctx, evc), ev_c1 = next_event(ctx, evc), ev_end = next_event(ctx, evc);
  SDF_HIP(hipEventRecord(ev_c0, st));
  unsigned long long total = 0;
  if (want_cigar) {
    {
      const int nb = (int)((n + 1023) / 1024);
      unsigned long long *d_part = d_total + 32;
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
    if (np)
      hipLaunchKernelGGL(cigar_compact_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, st, d_plan,
                         (int)np, d_out, d_stage, d_cig, (unsigned long long)cigar_cap);
  }
  SDF_HIP(hipEventRecord(ev_c1, st));
  SDF_HIP(hipEventRecord(ev_end, st));
  SDF_HIP(hipStreamSynchronize(st));
  SDF_HIP(hipGetLastError());
  // DP / traceback time = length of the union of the chunks' intervals (chunks overlap when pipelined);
  // ms[6] = sum of the chunks' DP intervals (what a kernel trace adds up)
  {
    auto span = [&](bool tb, float &sum) {
      std::vector<std::pair<float, float>> iv;
      sum = 0.f;
      auto add = [&](hipEvent_t e0, hipEvent_t e1) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, ev_begin, e0);
        (void)hipEventElapsedTime(&b, ev_begin, e1);
        iv.push_back({a, b});
        sum += b - a;
      };
      for (auto &ev : cev) {
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
    float s0 = 0, s1 = 0;
    ctx->ms[0] = span(false, s0);
    ctx->ms[1] = span(true, s1);
    ctx->ms[6] = s0;
  }
  (void)hipEventElapsedTime(&ctx->ms[2], ev_c0, ev_c1);
  (void)hipEventElapsedTime(&ctx->ms[3], ev_begin, ev_end);
  ctx->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  return SDF_OK;
}

extern "C" int sdf_extz2_batch(sdf_ctx *ctx, const sdf_scoring *sc, const sdf_task *tasks, size_t n,
                               const uint8_t *seq_pool, size_t pool_bytes, uint32_t want,
                               sdf_result *out, uint32_t *cigar_pool, size_t cigar_cap,
                               size_t *cigar_used) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (cigar_used) *cigar_used = 0;
  if (n == 0) return SDF_OK;
  if (!tasks || !out || (!seq_pool && pool_bytes)) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  SDF_HIP(hipSetDevice(ctx->device));
  // pack every referenced sequence once (2-bit codes + N mask) and rewrite offsets to words
  static const bool dbg_t = getenv("SDF_DEBUG_TIMING") != nullptr;
  const auto dbg0 = std::chrono::steady_clock::now();
  std::vector<sdf_task> t2(tasks, tasks + n);
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
      if (tasks[k].qlen > 0) sdf_pack_codes(seq_pool + tasks[k].q_off, tasks[k].qlen, packed + t2[k].q_off);
      if (tasks[k].tlen > 0) sdf_pack_codes(seq_pool + tasks[k].t_off, tasks[k].tlen, packed + t2[k].t_off);
    }
  };
  const int nthr = words >= (1u << 18) ? (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
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
  SDF_HIP(ctx->h_out.reserve(n * sizeof(sdf_result)));
  SDF_HIP(ctx->h_cig.reserve(std::max<size_t>(cigar_cap, 1) * 4));
  SDF_HIP(hipMemcpyAsync(ctx->h_pool.p, packed, std::max<size_t>(words, 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  size_t used = 0;
  int rc = sdf_extz2_batch_device(ctx, sc, t2.data(), n, (const uint32_t *)ctx->h_pool.p, want,
                                  (sdf_result *)ctx->h_out.p, (uint32_t *)ctx->h_cig.p, cigar_cap, &used,
                                  ctx->stream);
  if (cigar_used) *cigar_used = used;
  if (rc != SDF_OK) return rc;
  const auto dbg2 = std::chrono::steady_clock::now();
  const size_t out_bytes = n * sizeof(sdf_result), cig_bytes = (used && cigar_pool) ? used * 4 : 0;
  SDF_HIP(ctx->host_out.reserve(out_bytes + cig_bytes + 64));
  uint8_t *stg = (uint8_t *)ctx->host_out.p;
  SDF_HIP(hipMemcpyAsync(stg, ctx->h_out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (cig_bytes) SDF_HIP(hipMemcpyAsync(stg + out_bytes, ctx->h_cig.p, cig_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDF_HIP(hipStreamSynchronize(ctx->stream));
  {
    auto copy_range = [&](int q, int of) {
      const size_t a = out_bytes * (size_t)q / (size_t)of, b = out_bytes * (size_t)(q + 1) / (size_t)of;
      memcpy((uint8_t *)out + a, stg + a, b - a);
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
    fprintf(stderr, "[sdf_extz2_batch n=%zu words=%zu cap=%zu used=%zu] pack %.1f ms, h2d+device %.1f ms (plan %.1f, dp %.1f, tb %.1f), d2h %.1f ms\n",
            n, words, cigar_cap, used, ms(dbg0, dbg1), ms(dbg1, dbg2), ctx->ms[4], ctx->ms[0], ctx->ms[1], ms(dbg2, dbg3));
  }
  return SDF_OK;
}

// ---- seed anchors (reference: src/chain.cc:24-101) ---------------------------------------------------
static int anchors_range(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *d_pool, int kmer,
                         sdf_anchor *out, size_t out_cap, int64_t *out_off, size_t *out_used, hipStream_t st) {
  using namespace sdf;
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
  SDF_HIP(hipMemcpyAsync(d_pairs, hp.data(), n * sizeof(AnchorPairDev), hipMemcpyHostToDevice, st));
  const dim3 grid(32, (unsigned)n);
  hipLaunchKernelGGL(ref_keys_kernel, grid, dim3(256), 0, st, d_pairs, d_pool, kmer, d_keys);
  size_t tmp_bytes = 0;
  SDF_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys, d_keys2, (int)nrk, 0, 64, st));
  size_t scan_bytes = 0;
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (unsigned long long *)nullptr,
                                           (int)(nqk + 1), st));
  SDF_HIP(ctx->an_tmp.reserve(std::max(tmp_bytes, scan_bytes) + 256));
  SDF_HIP(hipcub::DeviceRadixSort::SortKeys(ctx->an_tmp.p, tmp_bytes, d_keys, d_keys2, (int)nrk, 0, 64, st));
  hipLaunchKernelGGL(query_lookup_kernel, grid, dim3(256), 0, st, d_pairs, d_pool, kmer, d_keys2, nrk, d_qlo, d_qcnt,
                     d_qeff, d_qpair);
  // exclusive scan over nqk+1 entries (the extra input element is ignored by the exclusive form)
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->an_tmp.p, scan_bytes, d_qeff, d_off, (int)(nqk + 1), st));
  unsigned long long ncand = 0;
  SDF_HIP(hipMemcpyAsync(&ncand, d_off + nqk, 8, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
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
                     d_qpair, nqk, (long long)ncand, d_flag, d_cand);
  SDF_HIP(hipMemsetAsync(d_flag + ncand, 0, 4, st));
  size_t scan2 = 0;
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan2, d_flag, d_pos, (int)(ncand + 1), st));
  SDF_HIP(ctx->an_tmp.reserve(scan2 + 256));
  SDF_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->an_tmp.p, scan2, d_flag, d_pos, (int)(ncand + 1), st));
  unsigned long long total = 0;
  SDF_HIP(hipMemcpyAsync(&total, d_pos + ncand, 8, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
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
    SDF_HIP(ctx->an_out.reserve((size_t)total * sizeof(CandOut)));
    hipLaunchKernelGGL(anchors_compact_kernel, dim3(nb), dim3(256), 0, st, d_flag, d_pos, d_cand, (long long)ncand,
                       (CandOut *)ctx->an_out.p, total);
    SDF_HIP(hipMemcpyAsync(out, ctx->an_out.p, (size_t)total * sizeof(sdf_anchor), hipMemcpyDeviceToHost, st));
  }
  SDF_HIP(hipStreamSynchronize(st));
  SDF_HIP(hipGetLastError());
  return SDF_OK;
}

extern "C" int sdf_anchors_batch(sdf_ctx *ctx, const sdf_anchor_pair *pairs, size_t n, const char *seq_pool,
                                 size_t pool_bytes, int kmer, sdf_anchor *out, size_t out_cap, int64_t *out_off,
                                 size_t *out_used) {
  if (!ctx) return SDF_ERR_INVALID;
  ctx->err.clear();
  if (out_used) *out_used = 0;
  if (!pairs || !out_off || !out_used || (!seq_pool && pool_bytes) || n >= (1u << 20)) {
    ctx->err = "invalid arguments";
    return SDF_ERR_INVALID;
  }
  if (kmer < 1 || kmer > 11) {
    ctx->err = "GPU anchors implement k-mer sizes up to 11";
    return SDF_ERR_UNSUPPORTED;
  }
  for (size_t i = 0; i < n; i++) {
    const sdf_anchor_pair &p = pairs[i];
    if (p.qlen < 0 || p.rlen < 0 || p.qlen >= (1 << 22) || p.rlen >= (1 << 22)) {
      ctx->err = "GPU anchors implement sequences shorter than 4 Mb";
      return SDF_ERR_UNSUPPORTED;
    }
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
  static const bool dbg_t = getenv("SDF_DEBUG_TIMING") != nullptr;
  const auto dbg0 = std::chrono::steady_clock::now();
  SDF_HIP(ctx->an_pool.reserve(pool_bytes + 16));
  SDF_HIP(hipMemcpyAsync(ctx->an_pool.p, seq_pool, pool_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (dbg_t) SDF_HIP(hipStreamSynchronize(ctx->stream));
  const auto dbg1 = std::chrono::steady_clock::now();
  const int rc = anchors_range(ctx, pairs, n, (const char *)ctx->an_pool.p, kmer, out, out_cap, out_off, out_used, ctx->stream);
  if (dbg_t)
    fprintf(stderr, "[sdf_anchors_batch n=%zu pool=%zu anchors=%zu] upload %.1f ms, rest %.1f ms\n", n, pool_bytes, *out_used,
            std::chrono::duration<double, std::milli>(dbg1 - dbg0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg1).count());
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
  std::vector<int64_t> ws_off(n + 1);
  int64_t words = 0;
  for (size_t i = 0; i < n; i++) {
    const int64_t m = off[i + 1] - off[i];
    if (off[0] != 0 || m < 0 || m >= (1 << 26)) {
      ctx->err = "anchor offsets must start at 0, ascend, and hold fewer than 2^26 anchors per pair";
      return SDF_ERR_INVALID;
    }
    ws_off[i] = words;
    if (m > 0) {
      int bits = 0;
      for (unsigned v = (unsigned)m - 1u; v; v >>= 1) ++bits;
      words += 12 * m + 4 * ((int64_t)2 << bits);
    }
  }
  ws_off[n] = words;
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
  hipLaunchKernelGGL(sdf::chain_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st,
                     (const sdf_anchor *)ctx->ch_an.p, (const int64_t *)ctx->ch_off.p, (const int64_t *)ctx->ch_wsoff.p,
                     (int)n, max_chain_gap, match_chain_score, (int32_t *)ctx->ch_work.p, (int32_t *)ctx->ch_path.p,
                     (int32_t *)ctx->ch_bounds.p, (int32_t *)ctx->ch_nb.p);
  SDF_HIP(hipGetLastError());
  if (total) SDF_HIP(hipMemcpyAsync(path, ctx->ch_path.p, total * 4, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipMemcpyAsync(bounds, ctx->ch_bounds.p, (total + n) * 8, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipMemcpyAsync(nbound, ctx->ch_nb.p, n * 4, hipMemcpyDeviceToHost, st));
  SDF_HIP(hipStreamSynchronize(st));
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

// ---- debugging aid (not part of the public header): copy the head of the direction workspace ----
extern "C" int sdf_debug_copy_dir(sdf_ctx *ctx, void *host, size_t bytes) {
  if (!ctx || !ctx->dir_ws.p) return SDF_ERR_INVALID;
  if (bytes > ctx->dir_ws.cap) bytes = ctx->dir_ws.cap;
  return hipMemcpy(host, ctx->dir_ws.p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? SDF_OK : SDF_ERR_HIP;
}
