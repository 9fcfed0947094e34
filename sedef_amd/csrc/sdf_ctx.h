// Context, device / pinned buffers and kernel declarations shared by the pieces of the C-ABI layer
// (sdf_plan.hip: batch cutting and chunk planning; sdf_launch.hip: uploads and launches; sdf_api.hip: entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
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
template <int NREG, bool STREAM, bool TRACK>
__global__ void extz2_pair_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                  sdf_result *);
size_t pair_lds_bytes(int qlen, int tlen, int nreg);
bool pair_fits_whole(int qlen, int tlen, int nreg);
size_t pair_mixed_lds_bytes(int qmax, int tmax, int nreg);
constexpr int kMixedMaxNeed = 576;  // widest window of a mixed pair: nine registers of 64 slots (w = 512)
template <int NREG>
__global__ void extz2_stripe_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *,
                                    sdf_result *, int, unsigned long long *, int, unsigned *);
__global__ void stripe_sync_init_kernel(const PlanTask *, const int32_t *, int, uint8_t *);
template <int NREG>
__global__ void extz2_bstripe_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *, sdf_result *,
                                     unsigned long long *, int, unsigned *);
__global__ void bstripe_init_kernel(const PlanTask *, const int32_t *, int, uint8_t *);
__global__ void bstripe_finish_kernel(const PlanTask *, const int32_t *, int, int, const uint8_t *, sdf_result *);
struct LaneRec;
__global__ void lane_keys_kernel(const LaneRec *, int, uint32_t *, uint32_t *);
__global__ void lane_sizes_kernel(const LaneRec *, const uint32_t *, int, unsigned long long *, unsigned long long *);
__global__ void lane_plan_kernel(const LaneRec *, const uint32_t *, int, const unsigned long long *, const unsigned long long *,
                                 int64_t, int64_t, PlanTask *);
__global__ void lane_hist_kernel(const LaneRec *, int, uint32_t *);
__global__ void lane_bins_scan_kernel(const uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, uint32_t *,
                                      unsigned long long *, unsigned long long *);
__global__ void lane_bins_top_kernel(uint32_t *, unsigned long long *, unsigned long long *);
__global__ void lane_place_kernel(const LaneRec *, int, uint32_t *, const uint32_t *, const unsigned long long *,
                                  const unsigned long long *, const uint32_t *, const unsigned long long *,
                                  const unsigned long long *, int64_t, int64_t, PlanTask *);
__global__ void extz2_lane_kernel(const PlanTask *, int, const uint32_t *, ScoreK, uint8_t *, sdf_result *);
__global__ void extz2_strip_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *, sdf_result *);
template <int C>
__global__ void extz2_strip_chain_kernel(const PlanTask *, const int32_t *, const uint32_t *, ScoreK, uint8_t *, sdf_result *,
                                         unsigned long long *, int, unsigned *);
__global__ void strip_chain_init_kernel(const PlanTask *, const int32_t *, uint8_t *);
template <int LAYOUT, int G>
__global__ void traceback_kernel(const PlanTask *, int, const uint32_t *, const uint8_t *, sdf_result *, uint32_t *);
__global__ void cigar_scan_blocks_kernel(sdf_result *, int, unsigned long long *);
__global__ void cigar_scan_parts_kernel(unsigned long long *, int, unsigned long long *);
__global__ void cigar_scan_add_kernel(sdf_result *, int, const unsigned long long *);
__global__ void cigar_compact_kernel(const PlanTask *, int, const sdf_result *, const uint32_t *,
                                     uint32_t *, unsigned long long);

__global__ void reset_results_kernel(sdf_result *res, int n);
struct PackRec;
__global__ void pack_chars_kernel(const PackRec *, long long, const char *, uint32_t *);
// (diagnostics of buffers that have no context to ask: set by every sdf_create from its configuration's debug_timing)
inline std::atomic<bool> g_debug_timing{false};

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  // `limit`: no headroom beyond this many bytes (the direction-flag workspace: the context's budget)
  hipError_t reserve(size_t bytes, size_t limit = ~(size_t)0) {
    if (bytes <= cap) return hipSuccess;
    const bool dbg_t = g_debug_timing.load(std::memory_order_relaxed);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t old = cap;
    // hipFree waits for the whole DEVICE -- the other lanes' batches included: 100-170 ms measured in a stage run, where
    // the allocation itself takes 0.2 ms -- so an outgrown buffer is only retired here: work of THIS call may still use it.
    // Buffers retired during EARLIER calls (new_call() has moved them to `stale`) are idle and are freed now, in the stall
    // this growth costs anyway: a context holds its buffer and at most the one it outgrew last (every outgrown
    // buffer used to stay until the context died).
    release_stale();
    if (p) {
      retired.push_back(p);
      retired_bytes += cap;
    }
    p = nullptr;
    cap = 0;
    size_t want = bytes + std::min<size_t>(bytes / 2, (size_t)8 << 30) + 4096;
    if (want > limit) want = std::max(bytes, limit);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {  // short of memory: give the retired buffers back first, then without headroom
      (void)hipGetLastError();
      for (void *q : retired) (void)hipFree(q);
      retired.clear();
      retired_bytes = 0;
      e = hipMalloc(&p, want);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        e = hipMalloc(&p, want);
      }
    }
    if (e == hipSuccess) cap = want;
    if (dbg_t && want >= (64u << 20))
      fprintf(stderr, "[DevBuf %zu -> %zu MiB in %.1f ms]\n", old >> 20, want >> 20,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return e;
  }
  // without headroom (sdf_reserve: the caller's bound IS the headroom)
  hipError_t reserve_exact(size_t bytes) { return reserve(bytes, bytes); }
  // at the start of a batch call (nothing of the context's earlier calls is in flight): what earlier calls retired is idle
  // from here on -- freed at the next growth, or right away when it is more than `keep` bytes
  void new_call(size_t keep = (size_t)4 << 30) {
    stale.insert(stale.end(), retired.begin(), retired.end());
    stale_bytes += retired_bytes;
    retired.clear();
    retired_bytes = 0;
    if (stale_bytes > keep) release_stale();
  }
  void release_stale() {
    for (void *q : stale) (void)hipFree(q);
    stale.clear();
    stale_bytes = 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    for (void *q : retired) (void)hipFree(q);
    retired.clear();
    retired_bytes = 0;
    release_stale();
    p = nullptr;
    cap = 0;
  }
  size_t held_bytes() const { return cap + retired_bytes + stale_bytes; }
  std::vector<void *> retired, stale;
  size_t retired_bytes = 0, stale_bytes = 0;
};

struct HostBuf {  // pinned host memory
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    retire();  // (hipHostFree waits for the device like hipFree: freed with the context)
    // (pinning is slow -- about a millisecond per 4 MB -- and a stage run sees its batches grow: half as much again as
    // headroom, so that a context re-pins a few times, not for every batch)
    const size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  hipError_t reserve_exact(size_t bytes) {  // (sdf_reserve: no headroom on top of the caller's bound)
    if (bytes <= cap) return hipSuccess;
    retire();
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  // Large staging the HOST fills (a super-batch's characters): memory of the process's own on transparent huge pages,
  // registered with the runtime.  Measured (profiles/pin_probe.py, 182 MB): hipHostMalloc 27-30 ms + 15-22 ms to free;
  // aligned_alloc(2 MB) + MADV_HUGEPAGE + hipHostRegister 0.5 ms once the pages exist, and the pages come with the first
  // write of the threads that fill them (11 ms of zero-fill for all of it against 33-36 ms on 4 KB pages); uploads from it
  // run at the same 55 GB/s.  Falls back to hipHostMalloc where registering fails.
  hipError_t reserve_huge(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    retire();
    const size_t two_mb = (size_t)2 << 20, want = (bytes + two_mb - 1) / two_mb * two_mb;
    void *q = aligned_alloc(two_mb, want);
    if (q) {
      (void)madvise(q, want, MADV_HUGEPAGE);
      if (hipHostRegister(q, want, hipHostRegisterDefault) == hipSuccess) {
        p = q;
        cap = want;
        registered = true;
        return hipSuccess;
      }
      (void)hipGetLastError();
      free(q);
    }
    return reserve_exact(bytes);
  }
  hipError_t reserve_pinned(bool registered_huge, size_t bytes) { return registered_huge ? reserve_huge(bytes) : reserve_exact(bytes); }
  void retire() {  // the buffer in use is outgrown: kept until the context goes (each kind on its own list)
    if (p) (registered ? retired_reg : retired).push_back(p);
    p = nullptr;
    cap = 0;
    registered = false;
  }
  void release() {
    if (p) {
      if (registered) {
        (void)hipHostUnregister(p);
        free(p);
      } else {
        (void)hipHostFree(p);
      }
    }
    for (void *q : retired) (void)hipHostFree(q);
    for (void *q : retired_reg) {
      (void)hipHostUnregister(q);
      free(q);
    }
    retired.clear();
    retired_reg.clear();
    p = nullptr;
    cap = 0;
    registered = false;
  }
  std::vector<void *> retired, retired_reg;
  bool registered = false;
};

}  // namespace sdf

namespace sdf {
struct BatchCut;

// A few parked host threads for the planning of a batch (a thread costs ~0.1 ms to start and join, a planning pass over
// a hundred thousand tasks less than a millisecond: the threads are started once per context).
class WorkerPool {
 public:
  explicit WorkerPool(int n, int spin_us = 0) {
    spin_us_ = spin_us > 0 ? spin_us : 0;
    for (int t = 0; t < n; ++t) threads_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  int size() const { return (int)threads_.size(); }
  void submit(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> g(mu_);
      jobs_.push_back(std::move(job));
      ++pending_;
      njobs_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_one();
  }
  void wait_idle() {  // every submitted job has finished
    std::unique_lock<std::mutex> g(mu_);
    idle_.wait(g, [&] { return pending_ == 0; });
  }

 private:
  void loop() {
    for (;;) {
      std::function<void()> job;
      {
        // A parked thread takes 0.1 - 3 ms to wake up on a box with a CPU quota -- as long as the whole scan of a million
        // tasks it is woken for.  SDF_POOL_SPIN_US lets a thread keep looking for the next job for a while before it
        // parks.  Off by default: measured on the 40,000-pair stage run (three lanes, 16-CPU quota) 300 us of looking cost
        // more in throttled periods than it saved -- 0.42-0.95 s per run against 0.32-0.46 s.
        const auto t0 = std::chrono::steady_clock::now();
        bool got = false;
        while (!got && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us_)) {
          if (njobs_.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> g(mu_);
            if (!jobs_.empty()) {
              job = std::move(jobs_.front());
              jobs_.pop_front();
              njobs_.fetch_sub(1, std::memory_order_release);
              got = true;
            }
          } else {
            __builtin_ia32_pause();
          }
        }
        if (!got) {
          std::unique_lock<std::mutex> g(mu_);
          cv_.wait(g, [&] { return quit_ || !jobs_.empty(); });
          if (jobs_.empty()) return;
          job = std::move(jobs_.front());
          jobs_.pop_front();
          njobs_.fetch_sub(1, std::memory_order_release);
        }
      }
      job();
      {
        std::lock_guard<std::mutex> g(mu_);
        if (--pending_ == 0) idle_.notify_all();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::deque<std::function<void()>> jobs_;
  std::mutex mu_;
  std::condition_variable cv_, idle_;
  int pending_ = 0;
  bool quit_ = false;
  std::atomic<int> njobs_{0};
  int spin_us_ = 0;  // sdf_config.pool_spin_us (set_spin_us)
};
}
using sdf::DevBuf;
using sdf::HostBuf;

struct sdf_ctx {
  sdf_config cfg;  // the context's settings, fixed when it is made (sdf_config.hip; sdf_api.hip: apply_config)
  int device = 0;
  hipStream_t stream = nullptr;
  size_t ws_budget = 0;
  hipStream_t dp_stream[2] = {nullptr, nullptr}, tb_stream = nullptr;  // chunk pipeline
  hipStream_t wide_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // chunks of several mixed-pair launches (sdf_launch.hip)
  size_t aux_limit = 4;  // extra pipeline streams this context may create (sdf_reserve with SDF_RESERVE_FEW_STREAMS: 0)
  hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};    // more room for launches that end in a tail
  DevBuf dir_ws, stage_ws, plan_buf, order_buf, misc_buf, gstate_buf;
  HostBuf host_plan, host_order;  // pinned staging of the plan
  HostBuf host_pool, host_out;    // pinned staging of the host-buffer entry point (packed sequences; results + CIGARs)
  DevBuf an_pool, an_pairs, an_keys, an_keys2, an_q, an_off, an_flag, an_pos, an_cand, an_out, an_tmp, an_outoff;
  DevBuf ch_an, ch_off, ch_wsoff, ch_work, ch_path, ch_bounds, ch_nb, ch_which;
  DevBuf st_tasks, st_pool, st_cig, st_out;  // sdf_stats_columns_batch
  DevBuf claim_buf;                          // stripe launches: eight entry counters each (stripe_claim), zeroed per call
  DevBuf st_items;                           // sdf_stats_columns_device: segments of long alignments + their counter
  unsigned stats_items = 1u << 18;           // ... capacity of that list (SDF_STATS_ITEMS)
  DevBuf h_pool, h_out, h_cig;  // device buffers of the host-buffer entry point
  DevBuf h_brief;               // ... 16-byte result records (sdf_extz2_batch_brief)
  std::vector<sdf_task> host_tasks;  // ... the task array with word offsets
  // lane kernel (extz2_lane.hip): records as uploaded, sort keys / values (in, out), sizes and their scans, hipCUB scratch
  HostBuf host_lane;
  HostBuf host_chars;      // pinned staging of a super-batch's FASTA characters (sdf_pool_host)
  size_t pool_bytes = 0;   // characters resident in an_pool (sdf_pool_upload / sdf_anchors_batch): what sdf_extz2_batch_pairs may name
  DevBuf pk_recs;          // ... one PackRec per task of such a call (seq_pack.hip)
  HostBuf host_an;  // pinned staging of the anchors call's output (sdf_reserve with SDF_RESERVE_ANCHORS; a pageable copy runs at ~3 GB/s)
  DevBuf ln_recs, ln_keys, ln_vals, ln_sizes, ln_tmp;
  DevBuf ln_bins;  // ... second form of the lane planning: per-key counts, ranks, prefixes (extz2_lane.hip: lane_hist_kernel and on)
  hipStream_t lane_stream = nullptr;
  bool strip_enabled = true;  // SDF_NO_STRIP=1: full-band tasks of 257..8192 target bases stay on the window / stripe kernels
  bool strip_always = false;  // SDF_STRIP_ALWAYS=1 (tests): the strip kernels whatever the number of tasks
  int strip_cols = 0;         // SDF_STRIP_COLS=4|8 (tests)
  bool lane_enabled = true;   // SDF_NO_LANE=1: small full-band tasks stay on the window kernels
  size_t lane_min = 8192;     // SDF_LANE_MIN: eligible tasks a batch must hold for the lane kernel to take them
  long long lane_tasks = 0;   // tasks of the last batch call the lane kernel took
  sdf::WorkerPool *pool = nullptr;  // planning threads, started with the first batch large enough to use them
  sdf::BatchCut *cut = nullptr;  // chunk list and planning scratch of the last batch call (sdf_plan.hip)
  std::vector<hipEvent_t> events;
  float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0 DP, 1 traceback, 2 compaction, 3 stream total, 4 host planning before the
                                           // first launch, 5 host total, 6 sum of the chunks' DP intervals
  int launches = 0;
  long long paired = 0;  // tasks of the last batch that ran two per wavefront (extz2_pair.hip)
  std::string err;
  int max_dyn_lds = 64 * 1024;
  bool force_general = false;  // SDF_FORCE_GENERAL=1: route everything to the LDS-resident kernel
  bool pipeline = true;        // SDF_PIPELINE=0: one chunk on one stream (isolated kernel timing)
  int stripe_min = 400;       // SDF_STRIPE_MIN: targets longer than this (and full band) take the stripe kernel
  int bstripe_min_rows = 4000;  // SDF_BSTRIPE_MIN_ROWS: banded tasks of this many anti-diagonals or more take the banded
                               // stripe kernel (extz2_bstripe.hip); 0: never
  bool no_stripe = false;      // SDF_NO_STRIPE=1: wide full-band tasks stay on the general kernel (extz2_stripe.hip off)
  size_t self_pair_max = 512;  // SDF_SELF_PAIR_MAX: see PlanEnv
  size_t chain_min = 3072;     // SDF_CHAIN_MIN: see PlanEnv
  bool stripe_claim = true;    // SDF_STRIPE_CLAIM=0: the stripe kernels' workgroups take launch-order entry blockIdx.x
  bool no_pair = false;        // SDF_NO_PAIR=1: never pack two tasks into one wavefront (extz2_pair.hip)
  bool no_mixed = false;       // SDF_NO_MIXED=1: no mixed pairs (banded tasks of different lengths in one wavefront)
  size_t mixed_min = 4096;     // SDF_MIXED_MIN: see PlanEnv
  int stripe_spin_cap = 1 << 24;  // SDF_STRIPE_SPIN_CAP: polls before a stripe's wait gives its task up (extz2_stripe.hip)
  sdf_ctx *part_ctx = nullptr;    // second context of this device: the first part of a very large batch (sdf_api.hip)
  hipEvent_t part_ev = nullptr;
  bool is_part = false, pool_shared = false;
  sdf_ctx *rerun_ctx = nullptr;   // context without stripe kernels for the tasks they gave up (created when first needed)
  DevBuf rr_out, rr_cig, rr_map;  // its outputs, and the (record, staging slot) map of the merge
  long long reran = 0;            // tasks of the last batch call that were re-run after a stripe gave up
};

// a context another context owns (the first part of a split batch, the re-run of abandoned stripe tasks): not another
// user of the process's CPUs (defined in sdf_api.hip)
void mark_internal_context(sdf_ctx *c);

#define SDF_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return SDF_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)
