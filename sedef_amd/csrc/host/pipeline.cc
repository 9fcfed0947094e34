// fast_align + refine_chains as a resumable per-pair job, the DP providers, and the stage driver
// (restates reference src/chain.cc:203-268, src/refine.cc:23-193, src/align_main.cc:200-337).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <fstream>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <thread>

#include "../../../include/sedef_hip.h"
#include <dirent.h>
#include <sys/stat.h>

#include "sedef_host.h"

namespace sdfh {

void set_alignment_scoring(const Params &p);
static void parallel_for(int n, const std::function<void(int)> &body);  // all host cores (defined below)

// ======================================================================================================
// DP providers: align_helper (src/align.cc:39-68) for a batch of requests
// ======================================================================================================
namespace {

struct TaskRef {
  size_t req;
  size_t q_off, t_off;
  int qlen, tlen;
};

// chunking of one request into DP tasks (src/align.cc:46-57): both sequences advance by the same SP.
// A round of the stage holds hundreds of thousands of requests (708,600 in the chr1-sized run): offsets and task counts in one
// serial pass, the copies and the task records on the host threads, the pool left unfilled where nothing is copied.
struct TaskPool {
  std::unique_ptr<uint8_t[]> bytes;
  size_t size = 0;
  const uint8_t *data() const { return bytes.get(); }
};
struct AlignCodes {  // align_dna (src/common.h:60-70,91): ACGT of either case 0..3, anything else the wildcard 4
  uint8_t of[256];
  AlignCodes() {
    for (int c = 0; c < 256; c++) {  // (the table is indexed with c & 127, like the host's align_dna)
      const int u = (c & 127) & ~0x20;
      of[c] = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
    }
  }
};
const AlignCodes kCodes;
void expand(const std::vector<DpRequest> &reqs, const Params &p, std::vector<TaskRef> &tasks, TaskPool &pool) {
  const size_t n = reqs.size(), step = (size_t)p.max_ksw_seq_len;
  std::vector<size_t> off(n + 1, 0), first(n + 1, 0);
  for (size_t k = 0; k < n; k++) {
    const DpRequest &r = reqs[k];
    off[k + 1] = off[k] + (size_t)r.qlen + (size_t)r.tlen;
    const size_t lim = (size_t)std::min(r.qlen, r.tlen);
    first[k + 1] = first[k] + (lim + step - 1) / step;
  }
  pool.size = off[n];
  pool.bytes.reset(new uint8_t[std::max<size_t>(off[n], 1)]);
  tasks.resize(first[n]);
  const size_t block = 4096;
  parallel_for((int)((n + block - 1) / block), [&](int b) {
    for (size_t k = (size_t)b * block; k < std::min(n, ((size_t)b + 1) * block); k++) {
      const DpRequest &r = reqs[k];
      const size_t qo = off[k], to = qo + (size_t)r.qlen;
      uint8_t *dq = pool.bytes.get() + qo, *dt = pool.bytes.get() + to;
      for (int i = 0; i < r.qlen; i++) dq[i] = kCodes.of[(unsigned char)r.q[i]];
      for (int i = 0; i < r.tlen; i++) dt[i] = kCodes.of[(unsigned char)r.t[i]];
      const size_t lim = (size_t)std::min(r.qlen, r.tlen);
      size_t at = first[k];
      for (size_t sp = 0; sp < lim; sp += step) {
        TaskRef &t = tasks[at++];
        t.req = k;
        t.q_off = qo + sp;
        t.t_off = to + sp;
        t.qlen = (int)std::min<size_t>(step, (size_t)r.qlen - sp);
        t.tlen = (int)std::min<size_t>(step, (size_t)r.tlen - sp);
      }
    }
  });
}

inline void append_ops(Cigar &c, const uint32_t *w, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    const int idx = w[i] & 0xf, len = (int)(w[i] >> 4);
    if (idx < 3) c.push_back({"MDI"[idx], len});  // ksw I (query) -> 'D', ksw D (target) -> 'I'
  }
}

void fill_mat(const Params &p, int8_t mat[25]) {  // src/align.cc:41-44
  const int8_t a = (int8_t)p.match, b = p.mismatch < 0 ? (int8_t)p.mismatch : (int8_t)(-p.mismatch);
  const int8_t m[25] = {a, b, b, b, 0, b, a, b, b, 0, b, b, a, b, 0, b, b, b, a, 0, 0, 0, 0, 0, 0};
  memcpy(mat, m, 25);
}

class GpuProvider : public DpProvider {
 public:
  // (`spares`: that many more providers, started before this one's own context so that all of them are set up side by side)
  // `max_batch_bytes` (0: unknown, prepare() will tell): the buffers are sized here, with the context.
  // `lanes`: the providers that share the device's memory with this one (itself included): the workspace budget is per process.
  explicit GpuProvider(int device, int spares = 0, const std::vector<int> &devices = std::vector<int>(),
                       size_t max_batch_bytes = 0, int lanes = 0)
      : device_(device), ws_(stage_workspace(lanes > 0 ? lanes : spares + 1)), lanes_(lanes > 0 ? lanes : spares + 1) {
    if (spares > 0) start_spares(spares, devices, max_batch_bytes);
    ctx_ = sdf_create(device, ws_);
    if (!ctx_) {
      for (auto &t : spare_threads_)
        if (t.joinable()) t.join();
      throw std::string("GPU DP backend unavailable: ") + sdf_last_error(nullptr);
    }
    if (max_batch_bytes) {
      prepared_ = true;
      reserve(max_batch_bytes);
    }
  }
  ~GpuProvider() override {
    ready();
    for (auto &t : spare_threads_)
      if (t.joinable()) t.join();
    {  // (the spare lanes' device contexts side by side: a context takes 30-40 ms to give back)
      std::vector<std::thread> gone;
      for (auto &e : spares_)
        if (e) gone.emplace_back([&e] { e.reset(); });
      for (auto &t : gone) t.join();
    }
    spares_.clear();
    sdf_destroy(ctx_);
  }
  // Direction-flag workspace of a stage lane: 16 GiB per process (round 6: the far-gap round of a chr1-sized bucket -- 10,813
  // tasks whose flag bound is 15 GB -- then runs as ONE chunk, one 7 ms chain instead of two: its DP 14.6 -> 10.2 ms,
  // profiles/r06_stage_dp2.txt; 8 GiB until then), shared out over its lanes (at least 2 GiB each), unless
  // SDF_STAGE_WS_GIB gives the figure per lane (the library's default is 64).  A round of the stage that needs more runs in
  // chunks; in exchange a lane allocates its workspace ONCE, where it is set up.  A process that starts right after another
  // one released tens of gigabytes sometimes waits SECONDS for a large hipMalloc (profiles/alloc_probe.py: 24 and 64 GiB
  // 0.3 ms or 1.8-4.0 s, 8 GiB 0.3 ms in every sample; inside a stage run: a 23 GiB request 0.4 or 580 ms) -- and a run of
  // `sedef align` is one such process per bucket, one after the other.  (Measured at the end of round 6 for 16 GiB as well:
  // the CLI asks for 8 when it is given ONE bucket, sedef_main.cc, profiles/r06_proc_probe.txt.)
  static size_t stage_workspace(int lanes) {
    const double asked = stage_settings().stage_ws_gib;
    const double gib = asked > 0 ? asked : std::max(2.0, 16.0 / std::max(lanes, 1));
    return (size_t)(gib * 1073741824.0);
  }
  // Spare providers for the other lanes, each created on a thread of its own (make_gpu_providers)
  void start_spares(int n, const std::vector<int> &devices, size_t max_batch_bytes) {
    spares_.resize((size_t)n);
    spare_dev_.resize((size_t)n);
    for (int i = 0; i < n; i++) {
      spare_dev_[(size_t)i] = devices.empty() ? device_ : devices[(size_t)(i + 1) % devices.size()];
      spare_threads_.emplace_back([this, i, n, max_batch_bytes] {
        try {
          spares_[(size_t)i].reset(new GpuProvider(spare_dev_[(size_t)i], 0, std::vector<int>(), max_batch_bytes, n + 1));
        } catch (std::string &) {  // (no room for another context: clone() will try again, the lane does without)
        }
      });
    }
  }
  std::unique_ptr<DpProvider> clone(int device = -1) override {
    const int want = device < 0 ? device_ : device;
    {
      std::lock_guard<std::mutex> g(spare_mu_);
      for (size_t i = 0; i < spares_.size(); i++) {
        if (spare_dev_[i] != want || spare_taken_.count(i)) continue;
        spare_taken_.insert(i);
        if (spare_threads_[i].joinable()) spare_threads_[i].join();
        if (spares_[i]) return std::unique_ptr<DpProvider>(spares_[i].release());
      }
    }
    return std::unique_ptr<DpProvider>(new GpuProvider(want, 0, std::vector<int>(), 0, 3));
  }
  void give_back(std::unique_ptr<DpProvider> p) override {
    GpuProvider *g = dynamic_cast<GpuProvider *>(p.get());
    if (!g) return;
    std::lock_guard<std::mutex> lk(spare_mu_);
    for (size_t i = 0; i < spares_.size(); i++)
      if (spare_taken_.count(i) && !spares_[i] && spare_dev_[i] == g->device_) {
        p.release();
        spares_[i].reset(g);
        spare_taken_.erase(i);
        return;
      }
    p.release();  // (made by clone() beyond the spares the provider started with: one more place)
    spares_.emplace_back(g);
    spare_dev_.push_back(g->device_);
    spare_threads_.emplace_back();
  }
  // Buffers sized once per lane (include/sedef_hip.h: sdf_reserve), on a thread of its own.  The bounds follow the stage's
  // rounds as measured: a task per ~250 bytes of a super-batch's sequences at most (chr1-sized run: 708,600 tasks of
  // 182 MB), a tenth of the bytes as bases of DP tasks.
  // (a provider that was not told the size when it was set up -- the C ABI's generate entry point -- sizes its buffers on a
  // thread of its own while the driver fetches sequences; pinning next to the first anchors upload slows that one down,
  // include/sedef_hip.h: sdf_reserve)
  void prepare(size_t max_batch_bytes) override {
    if (prepared_) return;
    prepared_ = true;
    reserve_thread_ = std::thread([this, max_batch_bytes] { reserve(max_batch_bytes); });
  }
  void reserve(size_t max_batch_bytes) {
    const size_t tasks = max_batch_bytes / 250 + 65536, bases = max_batch_bytes / 6 + (1u << 20);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = sdf_reserve(ctx_, tasks, bases, ws_, SDF_RESERVE_BRIEF | SDF_RESERVE_ANCHORS | (lanes_ > 1 ? SDF_RESERVE_FEW_STREAMS : 0u));
    // (the super-batch's characters are written straight into pinned memory and cross PCIe as one DMA: sized here, once)
    // (a sequence's slot is as long as its range in the FILE -- a line end per 50-80 bases --, and a pair's end may move to the
    // chromosome's: a sixteenth more than the bases, so that the stage never pins a second time)
    if (stage_settings().gpu_anchors) (void)sdf_pool_host(ctx_, max_batch_bytes + max_batch_bytes / 16 + (1u << 20));
    if (stage_settings().debug_timing)
      fprintf(stderr, "[sdf_reserve tasks %zu bases %zu: rc %d, %.1f ms]\n", tasks, bases, rc,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  void ready() {
    if (reserve_thread_.joinable()) reserve_thread_.join();
  }
  std::vector<Cigar> run(const std::vector<DpRequest> &reqs, const Params &p) override {
    std::vector<Cigar> out(reqs.size());
    Raw raw;
    run_raw(reqs, p, raw);
    const auto tp2 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < reqs.size(); r++) out[r] = raw.cigar(r);
    t_unpack += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count();
    return out;
  }

  bool run_raw(const std::vector<DpRequest> &reqs, const Params &p, Raw &raw) override {
    raw.first_task.assign(reqs.size() + 1, 0);
    raw.recs = nullptr;
    raw.words = nullptr;
    if (reqs.empty()) return true;
    std::vector<TaskRef> tr;
    TaskPool pool;
    const auto tp0 = std::chrono::steady_clock::now();
    expand(reqs, p, tr, pool);
    for (auto &t : tr) raw.first_task[t.req + 1]++;  // tasks are in request order
    for (size_t r = 0; r < reqs.size(); r++) raw.first_task[r + 1] += raw.first_task[r];
    if (tr.empty()) return true;
    const size_t nt = tr.size();
    std::unique_ptr<sdf_task[]> tasks(new sdf_task[nt]);
    const size_t tblock = 16384, ntb = (nt + tblock - 1) / tblock;
    parallel_for((int)ntb, [&](int b) {
      for (size_t k = (size_t)b * tblock; k < std::min(nt, ((size_t)b + 1) * tblock); k++)
        tasks[k] = make_task((int64_t)tr[k].q_off, (int64_t)tr[k].t_off, tr[k].qlen, tr[k].tlen);
    });
    call_batch(tasks.get(), nt, &pool, p, raw, tp0);
    return true;
  }

  // The same round on ranges of the character pool the last anchors() call left on the device: nothing is cut out, coded,
  // packed or uploaded here (include/sedef_hip.h: sdf_extz2_batch_pairs) -- a request becomes its 60 kb chunks
  // (src/align.cc:46-47: both sequences advance by the same offset) and that is all the host does per task.
  bool run_resident(const std::vector<ResidentReq> &reqs, const Params &p, Raw &raw) override {
    if (!resident_) return false;
    const size_t n = reqs.size(), step = (size_t)p.max_ksw_seq_len;
    raw.first_task.assign(n + 1, 0);
    raw.recs = nullptr;
    raw.words = nullptr;
    if (reqs.empty()) return true;
    const auto tp0 = std::chrono::steady_clock::now();
    const size_t rblock = 16384, nrb = (n + rblock - 1) / rblock;
    std::vector<size_t> bfirst(nrb + 1, 0);
    parallel_for((int)nrb, [&](int b) {
      size_t c = 0;
      for (size_t k = (size_t)b * rblock; k < std::min(n, ((size_t)b + 1) * rblock); k++)
        c += ((size_t)std::min(reqs[k].qlen, reqs[k].tlen) + step - 1) / step;
      bfirst[(size_t)b + 1] = c;
    });
    for (size_t b = 0; b < nrb; b++) bfirst[b + 1] += bfirst[b];
    const size_t nt = bfirst[nrb];
    raw.first_task[n] = nt;
    if (nt == 0) {
      std::fill(raw.first_task.begin(), raw.first_task.end(), 0);
      return true;
    }
    std::unique_ptr<sdf_task[]> tasks(new sdf_task[nt]);
    parallel_for((int)nrb, [&](int b) {
      size_t at = bfirst[(size_t)b];
      for (size_t k = (size_t)b * rblock; k < std::min(n, ((size_t)b + 1) * rblock); k++) {
        const ResidentReq &r = reqs[k];
        raw.first_task[k] = at;
        const size_t lim = (size_t)std::min(r.qlen, r.tlen);
        for (size_t sp = 0; sp < lim; sp += step)
          tasks[at++] = make_task(r.q_off + (int64_t)sp, r.t_off + (int64_t)sp, (int)std::min<size_t>(step, (size_t)r.qlen - sp),
                                  (int)std::min<size_t>(step, (size_t)r.tlen - sp));
      }
    });
    call_batch(tasks.get(), nt, nullptr, p, raw, tp0);
    return true;
  }

 private:
  static sdf_task make_task(int64_t q_off, int64_t t_off, int qlen, int tlen) {
    sdf_task t;
    memset(&t, 0, sizeof(t));
    t.q_off = q_off;
    t.t_off = t_off;
    t.qlen = qlen;
    t.tlen = tlen;
    t.w = -1;      // src/align.cc:86
    t.zdrop = -1;  // src/align.cc:54
    t.flag = 0;
    return t;
  }
  // one device batch call and its results as Raw: `pool` holds the tasks' codes (offsets are bytes of it), or is NULL when the
  // offsets are ranges of the resident character pool
  void call_batch(const sdf_task *tasks, size_t nt, const TaskPool *pool, const Params &p, Raw &raw,
                  std::chrono::steady_clock::time_point tp0) {
    const size_t tblock = 16384, ntb = (nt + tblock - 1) / tblock;
    std::vector<size_t> cap_part(ntb, 0);
    std::vector<int64_t> cells_part(ntb, 0);
    parallel_for((int)ntb, [&](int b) {
      size_t cap_b = 0;
      int64_t cells_b = 0;
      for (size_t k = (size_t)b * tblock; k < std::min(nt, ((size_t)b + 1) * tblock); k++) {
        cap_b += (size_t)tasks[k].qlen + tasks[k].tlen + 2;
        cells_b += (int64_t)tasks[k].qlen * tasks[k].tlen;
      }
      cap_part[(size_t)b] = cap_b;
      cells_part[(size_t)b] = cells_b;
    });
    size_t cap = 0;
    for (size_t b = 0; b < ntb; b++) cap += cap_part[b], cells += cells_part[b];
    this->tasks += (int64_t)nt;
    sdf_scoring sc;
    memset(&sc, 0, sizeof(sc));
    sc.m = 5;
    fill_mat(p, sc.mat);
    sc.gapo = (int8_t)(-p.gap_open);
    sc.gape = (int8_t)(-p.gap_extend);
    static_assert(sizeof(Raw::Rec) == sizeof(sdf_result_brief), "layouts must agree");
    size_t used = 0;
    ready();
    const auto tp1 = std::chrono::steady_clock::now();
    int rc;
    if (pool) {
      // (plain arrays, not std::vectors: the CIGAR capacity is the worst case, hundreds of megabytes that would be
      // zero-filled and paged in although the call writes only the words it reports in `used`)
      raw.own_recs.reset(new Raw::Rec[nt]);
      raw.own_words.reset(new uint32_t[cap]);
      rc = sdf_extz2_batch_brief(ctx_, &sc, tasks, nt, pool->data(), pool->size, (sdf_result_brief *)raw.own_recs.get(),
                                 raw.own_words.get(), cap, &used);
      raw.recs = raw.own_recs.get();
      raw.words = raw.own_words.get();
    } else {  // (read where the device's copies land: nothing is copied out of the pinned staging)
      const sdf_result_brief *res = nullptr;
      rc = sdf_extz2_batch_pairs_view(ctx_, &sc, tasks, nt, &res, &raw.words, &used);
      raw.recs = (const Raw::Rec *)res;
    }
    if (rc != SDF_OK) throw std::string("DP batch failed: ") + sdf_last_error(ctx_);
    const auto tp2 = std::chrono::steady_clock::now();
    t_pack += std::chrono::duration<double>(tp1 - tp0).count();
    t_call += std::chrono::duration<double>(tp2 - tp1).count();
  }

 public:
  char *pool_host(size_t bytes) override {
    if (!stage_settings().gpu_anchors) return nullptr;
    ready();
    uploaded_ = 0;
    pool_ = sdf_pool_host(ctx_, bytes + 64);
    pool_cap_ = pool_ ? bytes + 64 : 0;
    return pool_;
  }

  void pool_ready(size_t bytes) override {
    uploaded_ = 0;
    if (!pool_ || !stage_settings().gpu_anchors || bytes > pool_cap_) return;
    if (sdf_pool_upload(ctx_, pool_, bytes) == SDF_OK) uploaded_ = bytes;
  }

  // what the device kernels do not cover goes to the host's generate_anchors -- said once, not silently
  static bool host_instead(const char *why) {
    static std::atomic<bool> said(false);
    if (!said.exchange(true)) fprintf(stderr, "\n[sedef_amd] seed anchors on the host for this input: %s\n", why);
    return false;
  }
  // the pairs as the device takes them: offsets of the pool_host() buffer when the driver fetched the sequences into it
  // (`in_pool`), else back to back (the caller copies them); `total`: the pool's extent
  bool describe(const std::vector<AnchorJob> &jobs, std::vector<sdf_anchor_pair> &pairs, bool &in_pool, size_t &total) const {
    pairs.resize(jobs.size());
    total = 0;
    in_pool = pool_ != nullptr;
    for (size_t k = 0; k < jobs.size() && in_pool; k++)
      in_pool = jobs[k].query.data() >= pool_ && jobs[k].query.data() + jobs[k].query.size() <= pool_ + pool_cap_ &&
                jobs[k].ref.data() >= pool_ && jobs[k].ref.data() + jobs[k].ref.size() <= pool_ + pool_cap_;
    for (size_t k = 0; k < jobs.size(); k++) {
      if (jobs[k].query.size() >= (1u << 31) || jobs[k].ref.size() >= (1u << 31))
        return host_instead("GPU anchors implement sequences shorter than 2 Gb");
      pairs[k].q_off = in_pool ? (int64_t)(jobs[k].query.data() - pool_) : (int64_t)total;
      total += jobs[k].query.size();
      pairs[k].r_off = in_pool ? (int64_t)(jobs[k].ref.data() - pool_) : (int64_t)total;
      total += jobs[k].ref.size();
      pairs[k].qlen = (int32_t)jobs[k].query.size();
      pairs[k].rlen = (int32_t)jobs[k].ref.size();
      pairs[k].same_chr = jobs[k].same_chr;
      pairs[k].delta = jobs[k].delta;
    }
    if (in_pool) {  // (the pool's used extent, not the sum: slots are as long as the file's bytes, line ends included)
      total = 0;
      for (auto &pr : pairs) total = std::max<size_t>(total, (size_t)std::max(pr.q_off + pr.qlen, pr.r_off + pr.rlen));
    }
    return true;
  }
  void describe_out(const std::vector<sdf_anchor_pair> &pairs, AnchorBatch &out) const {
    out.resident = resident_;
    out.q_base.resize(pairs.size());
    out.r_base.resize(pairs.size());
    for (size_t k = 0; k < pairs.size(); k++) out.q_base[k] = pairs[k].q_off, out.r_base[k] = pairs[k].r_off;
  }

  // generate_anchors on the device (include/sedef_hip.h: sdf_anchors_batch)
  bool anchors(const std::vector<AnchorJob> &jobs, int kmer, AnchorBatch &out) override {
    if (!stage_settings().gpu_anchors || jobs.empty()) return false;
    ready();  // (a reserve still running on its thread uses the context: its warm-up call, its pinning next to this upload)
    if (kmer > 15) return host_instead("GPU anchors implement k-mer sizes up to 15");
    std::vector<sdf_anchor_pair> pairs;
    size_t total = 0;
    bool in_pool = false;
    if (!describe(jobs, pairs, in_pool, total)) return false;
    // The characters of all pairs in the context's pinned staging (sized with the lane's other buffers) -- where the driver
    // fetched them, or copied there now --, one asynchronous DMA, and they STAY on the device: the DP rounds of this
    // super-batch name their tasks as ranges of them (run_resident).
    resident_ = false;
    char *pool = pool_;
    if (!in_pool) {
      pool = pool_ = sdf_pool_host(ctx_, total + 1);
      pool_cap_ = pool ? total + 1 : 0;
      uploaded_ = 0;
      if (!pool) return host_instead(sdf_last_error(ctx_));
      parallel_for((int)jobs.size(), [&](int k) {
        memcpy(pool + pairs[k].q_off, jobs[k].query.data(), jobs[k].query.size());
        memcpy(pool + pairs[k].r_off, jobs[k].ref.data(), jobs[k].ref.size());
      });
    }
    const auto tu0 = std::chrono::steady_clock::now();
    const bool sent = in_pool && uploaded_ >= total && sdf_pool_bytes(ctx_) >= total;  // (pool_ready() has sent it)
    if (!sent) {
      if (sdf_pool_upload(ctx_, pool, total) != SDF_OK) return host_instead(sdf_last_error(ctx_));
      uploaded_ = total;
    }
    if (stage_settings().debug_timing)
      fprintf(stderr, "[anchors: %zu pairs, pool %zu bytes %s, upload enqueued in %.1f ms]\n", jobs.size(), total,
              in_pool ? "fetched in place" : "copied", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count());
    out.off.assign(jobs.size() + 1, 0);
    size_t used = 0;
    static_assert(sizeof(sdf_anchor) == sizeof(Anchor), "layouts must agree");
    // (the anchors stay in the context's pinned staging: the jobs copy theirs out when they start)
    const sdf_anchor *found = nullptr;
    out.buf.reset();
    int rc = sdf_anchors_batch_view(ctx_, pairs.data(), pairs.size(), nullptr, std::max<size_t>(total, uploaded_), kmer, &found,
                                    out.off.data(), &used);
    out.view = (const Anchor *)found;
    if (rc == SDF_ERR_UNSUPPORTED || rc == SDF_ERR_NOMEM) return host_instead(sdf_last_error(ctx_));
    if (rc != SDF_OK) throw std::string("GPU anchors failed: ") + sdf_last_error(ctx_);
    if (stage_settings().debug_timing)
      fprintf(stderr, "[anchors: device call returned %.1f ms after the upload was enqueued]\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count());
    resident_ = stage_settings().resident_dp;
    describe_out(pairs, out);
    return true;
  }

  bool anchors_more(const std::vector<AnchorJob> &jobs, int kmer, size_t keep, AnchorBatch &out) override {
    if (!stage_settings().gpu_anchors || jobs.empty() || kmer > 15 || !pool_ || !uploaded_) return false;
    std::vector<sdf_anchor_pair> pairs;
    size_t total = 0;
    bool in_pool = false;
    if (!describe(jobs, pairs, in_pool, total) || !in_pool || total > uploaded_) return false;
    out.off.assign(jobs.size() + 1, 0);
    size_t used = 0;
    const sdf_anchor *found = nullptr;
    out.buf.reset();
    const int rc = sdf_anchors_batch_more(ctx_, pairs.data(), pairs.size(), uploaded_, kmer, keep, &found, out.off.data(), &used);
    if (rc != SDF_OK) return false;  // (no room behind the kept anchors, or nothing the device covers: the caller's fallback)
    out.view = (const Anchor *)found;
    describe_out(pairs, out);
    return true;
  }

 private:
  sdf_ctx *ctx_;
  int device_;
  size_t ws_;
  int lanes_;
  int64_t tasks_ = 0;
  bool prepared_ = false;
  bool resident_ = false;  // the last anchors() call's characters are in HBM (sdf_pool_upload)
  char *pool_ = nullptr;   // the context's pinned character staging as last asked for (pool_host / anchors)
  size_t uploaded_ = 0;    // bytes of it on the device (pool_ready / anchors)
  size_t pool_cap_ = 0;
  std::thread reserve_thread_;
  std::vector<std::unique_ptr<GpuProvider>> spares_;
  std::vector<int> spare_dev_;
  std::vector<std::thread> spare_threads_;
  std::set<size_t> spare_taken_;
  std::mutex spare_mu_;
};

struct OracleResult {  // layout of sdfo_result (oracle/extz2_oracle.h)
  uint32_t max;
  int32_t zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score;
  int64_t n_cigar;
  uint32_t *cigar;
};

class TestProvider : public DpProvider {
 public:
  explicit TestProvider(test_dp_fn fn) : fn_(fn) {}
  std::vector<Cigar> run(const std::vector<DpRequest> &reqs, const Params &p) override {
    std::vector<Cigar> out(reqs.size());
    std::vector<TaskRef> tr;
    TaskPool pool;
    expand(reqs, p, tr, pool);
    int8_t mat[25];
    fill_mat(p, mat);
    // the hook is re-entrant (like ksw_extz2_sse): tasks run on all host threads, CIGARs are appended in task order
    std::vector<OracleResult> res(tr.size());
    parallel_for((int)tr.size(), [&](int k) {
      const TaskRef &t = tr[(size_t)k];
      memset(&res[(size_t)k], 0, sizeof(OracleResult));
      fn_(t.qlen, pool.data() + t.q_off, t.tlen, pool.data() + t.t_off, 5, mat, -p.gap_open, -p.gap_extend, -1, -1,
          0, &res[(size_t)k]);
    });
    for (size_t k = 0; k < tr.size(); k++) {
      append_ops(out[tr[k].req], res[k].cigar, res[k].n_cigar);
      free(res[k].cigar);
      tasks++;
      cells += (int64_t)tr[k].qlen * tr[k].tlen;
    }
    return out;
  }
  // (lanes of super-batches and of buckets work with the hook as with the device: the hook is re-entrant)
  std::unique_ptr<DpProvider> clone(int /*device*/ = -1) override { return std::unique_ptr<DpProvider>(new TestProvider(fn_)); }

 private:
  test_dp_fn fn_;
};

}  // namespace

Cigar DpProvider::Raw::cigar(size_t req) const {
  Cigar c;
  c.matches = 0;
  for (size_t k = first_task[req]; k < first_task[req + 1]; k++) {
    append_ops(c, words + recs[k].off, recs[k].cnt);
    c.matches += recs[k].match;
  }
  return c;
}

std::unique_ptr<DpProvider> make_gpu_provider(int device) { return std::unique_ptr<DpProvider>(new GpuProvider(device)); }

std::unique_ptr<DpProvider> make_gpu_providers(int device, int lanes, const std::vector<int> &devices, size_t max_batch_bytes) {
  return std::unique_ptr<DpProvider>(new GpuProvider(device, std::max(lanes - 1, 0), devices, max_batch_bytes));
}

static int stage_super_batch(int total, int nlanes, int super_batch) {
  if (nlanes > 1) super_batch = std::max(1024, std::min(super_batch, (total + 2 * nlanes - 1) / (2 * nlanes)));
  if (stage_settings().super_batch > 0) super_batch = stage_settings().super_batch;
  return super_batch;
}

StageHint stage_hint(const std::string &bed_path, int super_batch) {
  StageHint h;
  std::vector<int64_t> bytes;
  if (FILE *f = fopen(bed_path.c_str(), "rb")) {
    // columns 2, 3 and 5, 6 of a line: the two intervals (src/hit.cc: Hit::from_bed); anything else is skipped
    std::vector<char> line(1 << 16);
    while (fgets(line.data(), (int)line.size(), f)) {
      int64_t v[6] = {0, 0, 0, 0, 0, 0};
      int col = 0;
      for (const char *c = line.data(); *c && *c != '\n' && col < 6; ++col) {
        const char *e = c;
        while (*e && *e != '\t' && *e != '\n') ++e;
        if (col == 1 || col == 2 || col == 4 || col == 5) v[col] = atoll(c);
        c = *e == '\t' ? e + 1 : e;
      }
      bytes.push_back(std::max<int64_t>(0, v[2] - v[1]) + std::max<int64_t>(0, v[5] - v[4]));
    }
    fclose(f);
  }
  h.pairs = (int)bytes.size();
  h.lanes = stage_lane_count(h.pairs, &h.devices);
  h.super_batch = stage_super_batch(h.pairs, h.lanes, super_batch);
  // the largest super-batch holds at most the `super_batch` largest pairs (and at most 1 GiB + one pair)
  const size_t k = std::min<size_t>(bytes.size(), (size_t)h.super_batch);
  std::partial_sort(bytes.begin(), bytes.begin() + (long)k, bytes.end(), std::greater<int64_t>());
  int64_t sum = 0;
  for (size_t i = 0; i < k && sum < ((int64_t)1 << 30); i++) sum += bytes[i];
  h.max_batch_bytes = (size_t)sum;
  return h;
}

int stage_lane_count(int total, std::vector<int> *devices) {
  // A second context costs 50-120 ms to set up, so small inputs stay on one lane; medium ones are cut into at least two
  // super-batches per lane.  (measured: 40,000 pairs 1.19 / 0.93 / 1.03 s with 2 / 3 / 4 lanes)
  int nlanes = total >= 12288 ? 3 : total >= 4096 ? 2 : 1;
  // SDF_DEVICES=0,1,...: the lanes after the first go round-robin over these devices (one node, several GPUs; the
  // first lane stays on the provider's own device, which should be the first of the list); at least one lane each
  const std::vector<int> &dv = stage_settings().devices;
  if (dv.size() > 1) nlanes = std::max<int>(nlanes, (int)std::min<size_t>(dv.size(), 8));
  if (stage_settings().lanes > 0) nlanes = std::max(1, std::min(8, stage_settings().lanes));
  if (devices) *devices = dv;
  return nlanes;
}
std::unique_ptr<DpProvider> make_test_provider(test_dp_fn fn) { return std::unique_ptr<DpProvider>(new TestProvider(fn)); }

// ======================================================================================================
// PairJob
// ======================================================================================================
namespace {
std::atomic<long long> g_us_chain(0), g_us_rest(0);  // host CPU time (summed over threads), microseconds
struct ScopedUs {
  std::atomic<long long> &acc;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit ScopedUs(std::atomic<long long> &a) : acc(a) {}
  ~ScopedUs() { acc += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

struct PairJob::PathState {
  std::deque<int> idx;       // chain hits of the path, in order
  int qlo, qhi, rlo, rhi;    // extent from the chain coordinates (before any merge)
  bool est_ok = true;
  // progress
  size_t pi = 1;
  int prev = -1;             // index into hits_
  std::vector<Hit> guide;
  bool merging = false;      // waiting for the gap DP of a merge
  bool building = false;     // waiting for the DPs of the final guide alignment
  bool finished = false;
  Hit result;
};

PairJob::PairJob(SeqView query, SeqView ref, const Hit &orig, const Params &p)
    : query_(query), ref_(ref), orig_(orig), p_(p) {
  auto plain = [](SeqView s) {
    static const struct Tab {
      bool ok[256];
      Tab() {
        for (auto &b : ok) b = false;
        for (const char *c = "ACGTNacgtn"; *c; ++c) ok[(unsigned char)*c] = true;
      }
    } tab;
    for (unsigned char c : s)
      if (!tab.ok[c]) return false;
    return true;
  };
  exact_ = plain(query) && plain(ref);
}

void PairJob::stage_start(std::vector<DpRequest> &out) {  // src/chain.cc:203-258
  // (the hits' Sequence objects carry names and strands; nothing of the stage reads bases through them -- a copy of both
  // sequences per pair was 182 MB of memcpy and page faults in the chr1-sized run)
  query_ptr_ = std::make_shared<Sequence>("QRY", std::string());
  ref_ptr_ = std::make_shared<Sequence>("REF", std::string());
  if (!have_anchors_) anchors_ = generate_anchors(query_.str(), ref_.str(), orig_, p_.kmer);
  else anchors_.assign(ext_anchors_, ext_anchors_ + ext_count_);
  auto chains = chain_anchors(anchors_, p_);
  const auto &chain = chains.first;
  const auto &bounds = chains.second;
  const double min_span = p_.min_read_size * (1 - p_.max_error);
  for (size_t bi = 1; bi < bounds.size(); bi++) {
    const bool has_u = bounds[bi].second;
    const int be = bounds[bi].first, bs = bounds[bi - 1].first;
    const int up = bounds[bi].second;
    const int qlo = anchors_[chain[be - 1]].q, qhi = anchors_[chain[bs]].q + anchors_[chain[bs]].l;
    const int rlo = anchors_[chain[be - 1]].r, rhi = anchors_[chain[bs]].r + anchors_[chain[bs]].l;
    const int span = std::max(rhi - rlo, qhi - qlo);
    if ((!has_u || span < p_.min_uppercase_match) && span < min_span) continue;
    Hit a;
    a.query = query_ptr_;
    a.query_start = qlo;
    a.query_end = qhi;
    a.ref = ref_ptr_;
    a.ref_start = rlo;
    a.ref_end = rhi;
    a.jaccard = up;
    guides_.push_back(std::vector<int>());
    for (int k = be - 1; k >= bs; k--) guides_.back().push_back(chain[k]);
    hits_.push_back(a);
  }
  DpSession rec;
  rec.recording = true;
  rec.requests = &out;
  for (size_t k = 0; k < hits_.size(); k++) Alignment tmp(query_, ref_, anchors_, guides_[k], rec);
}

void PairJob::stage_chain_finish(const std::vector<Cigar> &results) {
  DpSession rep;
  rep.recording = false;
  rep.results = &results;
  rep.codes_are_exact = exact_;
  for (size_t k = 0; k < hits_.size(); k++) {
    hits_[k].aln = Alignment(query_, ref_, anchors_, guides_[k], rep);
    update_from_alignment(hits_[k]);
  }
  std::vector<Anchor>().swap(anchors_);
  plan_paths();
}

void PairJob::plan_paths() {  // src/refine.cc:23-147
  std::vector<Hit> &anchors = hits_;
  std::sort(anchors.begin(), anchors.end());
  const bool same_chr = orig_.query->name == orig_.ref->name && orig_.query->is_rc == orig_.ref->is_rc;
  std::vector<int> score;
  for (auto &a : anchors)
    score.push_back(p_.refine_match * a.aln.matches() - p_.refine_mismatch * a.aln.mismatches() -
                    p_.refine_gap * a.aln.gap_bases());  // double -> int
  std::vector<int> dp(anchors.size(), 0), prev(anchors.size(), -1);
  std::set<std::pair<int, int>, std::greater<std::pair<int, int>>> maxes;
  for (int ai = 0; ai < (int)anchors.size(); ai++) {
    if (same_chr) {
      auto &c = anchors[ai];
      const int qlo = c.query_start, qhi = c.query_end, rlo = c.ref_start, rhi = c.ref_end;
      const int qo = std::max(0, std::min(orig_.query_start + qhi, orig_.ref_start + rhi) -
                                     std::max(orig_.query_start + qlo, orig_.ref_start + rlo));
      if ((rhi - rlo) - qo < p_.refine_side_align && (qhi - qlo) - qo < p_.refine_side_align) continue;
    }
    dp[ai] = score[ai];
    for (int aj = ai - 1; aj >= 0; aj--) {
      auto &c = anchors[ai];
      auto &p = anchors[aj];
      int cqs = c.query_start;
      if (cqs < p.query_end) cqs = p.query_end;
      int crs = c.ref_start;
      if (crs < p.ref_end) crs = p.ref_end;
      if (p.query_end >= c.query_end || p.ref_end >= c.ref_end) continue;
      if (p.ref_start >= c.ref_start) continue;
      const int ma = std::max(cqs - p.query_end, crs - p.ref_end);
      const int mi = std::min(cqs - p.query_end, crs - p.ref_end);
      if (ma >= p_.refine_max_gap) continue;
      if (same_chr) {
        const int qlo = p.query_end, qhi = cqs, rlo = p.ref_end, rhi = crs;
        const int qo = std::max(0, std::min(orig_.query_start + qhi, orig_.ref_start + rhi) -
                                       std::max(orig_.query_start + qlo, orig_.ref_start + rlo));
        if (qo >= 1) continue;
      }
      const int mis = p_.refine_mismatch * mi, gap = p_.refine_gapopen + p_.refine_gap * (ma - mi);
      const int sco = dp[aj] + score[ai] - mis - gap;
      if (sco >= dp[ai]) {
        dp[ai] = sco;
        prev[ai] = aj;
      }
    }
    maxes.insert({dp[ai], ai});
  }
  std::vector<bool> used(anchors.size(), false);
  for (auto &m : maxes) {
    if (m.first == 0) break;
    int maxi = m.second;
    if (used[maxi]) continue;
    auto ps = std::make_shared<PathState>();
    while (maxi != -1 && !used[maxi]) {
      ps->idx.push_front(maxi);
      used[maxi] = true;
      maxi = prev[maxi];
    }
    ps->qlo = anchors[ps->idx.front()].query_start;
    ps->qhi = anchors[ps->idx.back()].query_end;
    ps->rlo = anchors[ps->idx.front()].ref_start;
    ps->rhi = anchors[ps->idx.back()].ref_end;
    int est_size = anchors[ps->idx[0]].aln.span();
    for (size_t i = 1; i < ps->idx.size(); i++) {
      est_size += anchors[ps->idx[i]].aln.span();
      est_size += std::max(anchors[ps->idx[i]].query_start - anchors[ps->idx[i - 1]].query_end,
                           anchors[ps->idx[i]].ref_start - anchors[ps->idx[i - 1]].ref_end);
    }
    ps->est_ok = est_size >= p_.refine_min_read - p_.refine_side_align;
    ps->prev = ps->idx[0];
    paths_.push_back(ps);
  }
}

std::vector<DpRequest> PairJob::advance(const std::vector<Cigar> &results) {
  std::vector<DpRequest> out;
  ScopedUs timer(stage_ == START ? g_us_chain : g_us_rest);
  if (stage_ == START) {
    stage_start(out);
    stage_ = CHAIN_ALN;
    if (!out.empty()) return out;
    // no DP needed: fall through with an empty result set
    stage_chain_finish(std::vector<Cigar>());
    stage_ = PATHS;
  } else if (stage_ == CHAIN_ALN) {
    stage_chain_finish(results);
    stage_ = PATHS;
  }
  if (stage_ != PATHS) return out;

  // distribute the results of the previous round to the paths that were waiting, in request order
  size_t cursor = 0;
  std::vector<int> waiting;
  waiting.swap(wait_paths_);
  std::vector<size_t> counts;
  counts.swap(wait_counts_);
  std::vector<std::vector<Cigar>> path_results(paths_.size());
  for (size_t w = 0; w < waiting.size(); w++) {
    path_results[waiting[w]].assign(results.begin() + cursor, results.begin() + cursor + counts[w]);
    cursor += counts[w];
  }

  const SeqView qseq = query_, rseq = ref_;
  for (size_t pk = 0; pk < paths_.size(); pk++) {
    PathState &ps = *paths_[pk];
    if (ps.finished || !ps.est_ok) continue;
    std::vector<Cigar> have;
    have.swap(path_results[pk]);
    bool have_results = ps.merging || ps.building;
    for (;;) {
      if (ps.building) {  // results of the final guide alignment are here
        DpSession rep;
        rep.recording = false;
        rep.results = &have;
        rep.codes_are_exact = exact_;
        ps.result.aln = Alignment(qseq, rseq, ps.guide, p_.refine_side_align, rep);
        update_from_alignment(ps.result);
        ps.finished = true;
        break;
      }
      if (ps.merging) {  // results of a merge's gap DP are here: do the real merge
        Hit &prev = hits_[ps.prev];
        Hit &cur = hits_[ps.idx[ps.pi]];
        DpSession rep;
        rep.recording = false;
        rep.results = &have;
        rep.codes_are_exact = exact_;
        prev.aln.merge(cur.aln, qseq, rseq, rep);
        update_from_alignment(prev);
        ps.merging = false;
        ps.pi++;
        have.clear();
        have_results = false;
      }
      // walk the path (src/refine.cc:165-179)
      bool waiting_now = false;
      while (ps.pi < ps.idx.size()) {
        Hit &prev = hits_[ps.prev];
        Hit &cur = hits_[ps.idx[ps.pi]];
        if (cur.query_start < prev.query_end || cur.ref_start < prev.ref_end) {
          // enumerate the merge's DP request on copies
          std::vector<DpRequest> reqs;
          DpSession rec;
          rec.recording = true;
          rec.requests = &reqs;
          Alignment pa = prev.aln, ca = cur.aln;
          pa.merge(ca, qseq, rseq, rec);
          if (reqs.empty()) {
            std::vector<Cigar> none;
            DpSession rep;
            rep.recording = false;
            rep.results = &none;
            rep.codes_are_exact = exact_;
            prev.aln.merge(cur.aln, qseq, rseq, rep);
            update_from_alignment(prev);
            ps.pi++;
            continue;
          }
          ps.merging = true;
          wait_paths_.push_back((int)pk);
          wait_counts_.push_back(reqs.size());
          out.insert(out.end(), reqs.begin(), reqs.end());
          waiting_now = true;
          break;
        }
        ps.guide.push_back(prev);
        ps.prev = ps.idx[ps.pi];
        ps.pi++;
      }
      if (waiting_now) break;
      // end of the path: the refined hit (src/refine.cc:163,180-183)
      ps.guide.push_back(hits_[ps.prev]);
      ps.result = Hit();
      ps.result.query = hits_.front().query;
      ps.result.query_start = ps.qlo;
      ps.result.query_end = ps.qhi;
      ps.result.ref = hits_.front().ref;
      ps.result.ref_start = ps.rlo;
      ps.result.ref_end = ps.rhi;
      std::vector<DpRequest> reqs;
      DpSession rec;
      rec.recording = true;
      rec.requests = &reqs;
      { Alignment tmp(qseq, rseq, ps.guide, p_.refine_side_align, rec); }
      ps.building = true;
      if (reqs.empty()) {
        have.clear();
        continue;  // build immediately
      }
      wait_paths_.push_back((int)pk);
      wait_counts_.push_back(reqs.size());
      out.insert(out.end(), reqs.begin(), reqs.end());
      break;
    }
    (void)have_results;
  }
  if (out.empty()) {
    finish_paths();
    stage_ = DONE;
  }
  return out;
}

void PairJob::finish_paths() {  // acceptance replay: src/refine.cc:143-162,184-190
  for (auto &pp : paths_) {
    PathState &ps = *pp;
    if (!ps.est_ok) continue;
    bool overlap = false;
    for (auto &h : final_hits_) {
      const int qo = std::max(0, std::min(ps.qhi, h.query_end) - std::max(ps.qlo, h.query_start));
      const int ro = std::max(0, std::min(ps.rhi, h.ref_end) - std::max(ps.rlo, h.ref_start));
      if (ps.qhi - ps.qlo - qo < p_.refine_side_align && ps.rhi - ps.rlo - ro < p_.refine_side_align) {
        overlap = true;
        break;
      }
    }
    if (overlap) continue;
    if (ps.result.aln.span() >= p_.refine_min_read) final_hits_.push_back(ps.result);
  }
  paths_.clear();
  hits_.clear();
}

// ======================================================================================================
// Stage driver
// ======================================================================================================
// The per-pair host work (anchors, chaining, refinement, CIGAR algebra) is independent across pairs: run it on
// all host cores (SDF_HOST_THREADS overrides).  The reference runs one single-threaded process per bucket.
StageSettings StageSettings::from_env() {
  StageSettings s;
  auto num = [](const char *name, long lo, long hi, long dflt) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end) throw std::string(name) + "=" + e + ": not a number";
    if (v < lo || v > hi) throw std::string(name) + "=" + e + ": out of range";
    return v;
  };
  s.device = (int)num("SDF_DEVICE", 0, 1023, 0);
  s.lanes = (int)num("SDF_LANES", 0, 8, 0);
  s.super_batch = (int)num("SDF_SUPER_BATCH", 0, 1 << 30, 0);
  s.gpu_anchors = num("SDF_GPU_ANCHORS", 0, 1, 1) != 0;
  s.host_threads = (int)num("SDF_HOST_THREADS", 0, 4096, 0);
  if (const char *e = getenv("SDF_STAGE_WS_GIB")) s.stage_ws_gib = atof(e) > 0 ? atof(e) : 0;
  s.debug_timing = getenv("SDF_DEBUG_TIMING") != nullptr;
  s.resident_dp = num("SDF_RESIDENT_DP", 0, 1, 1) != 0;
  s.anchor_parts = (int)num("SDF_ANCHOR_PARTS", 0, 16, 0);
  s.bucket_lanes = (int)num("SDF_BUCKET_LANES", 1, 4, 2);
  if (const char *e = getenv("SDF_DEVICES"))
    for (const char *c = e; *c;) {
      char *end = nullptr;
      const long d = strtol(c, &end, 10);
      if (end == c) throw std::string("SDF_DEVICES=") + e + ": a list of device ordinals";
      s.devices.push_back((int)d);
      c = *end == ',' ? end + 1 : end;
    }
  return s;
}
namespace {
StageSettings g_stage_settings;
std::mutex g_stage_settings_mu;
}  // namespace
const StageSettings &stage_settings() { return g_stage_settings; }
void set_stage_settings(const StageSettings &s) {
  std::lock_guard<std::mutex> g(g_stage_settings_mu);
  g_stage_settings = s;
}

static void parallel_for(int n, const std::function<void(int)> &body) {
  const int nthreads = [] {
    const int asked = stage_settings().host_threads;
    int t = asked > 0 ? asked : (int)std::thread::hardware_concurrency();
    if (asked <= 0) {  // a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>") counts, not the host's core count
      if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
          t = std::min<long long>(t, std::max<long long>(1, (quota + period - 1) / period));
        fclose(f);
      }
    }
    return t < 1 ? 1 : (t > 64 ? 64 : t);
  }();
  const int nt = std::min(nthreads, n);
  if (nt <= 1) {
    for (int i = 0; i < n; i++) body(i);
    return;
  }
  // The lanes of the stage driver call this concurrently, a few hundred times per run.  One set of worker threads for
  // the whole process (as many as the CPU quota: more runnable threads get the process throttled) serves all callers:
  // a call is a REGION -- a counter over its items -- queued for the workers; the caller works on its own region too,
  // so it never waits for a parked thread to wake up, and returns when all its items are done.  (Threads created and
  // joined per call cost 0.1 ms each, milliseconds under a CPU quota.)
  struct Region {
    const std::function<void(int)> *body;
    int n;
    std::atomic<int> next{0}, done{0};
    std::mutex err_mu;
    std::string error;
    void run() {
      for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
        try {
          (*body)(i);
        } catch (std::string &s) {
          std::lock_guard<std::mutex> g(err_mu);
          if (error.empty()) error = s.empty() ? std::string("error") : s;
        }
        if (done.fetch_add(1) + 1 == n) {  // the last item: wake the caller if it is waiting
          std::lock_guard<std::mutex> g(fin_mu);
          fin_cv.notify_all();
        }
      }
    }
    std::mutex fin_mu;
    std::condition_variable fin_cv;
  };
  struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Region>> open;  // regions that may still have items to hand out
    std::vector<std::thread> workers;
    bool quit = false;
    explicit Pool(int nw) {
      for (int t = 0; t < nw; t++)
        workers.emplace_back([this] {
          for (;;) {
            std::shared_ptr<Region> r;
            {
              std::unique_lock<std::mutex> g(mu);
              cv.wait(g, [&] {
                while (!open.empty() && open.front()->next.load() >= open.front()->n) open.pop_front();
                return quit || !open.empty();
              });
              if (open.empty()) return;
              r = open.front();  // (to the back: the next worker takes another caller's region -- the lanes share the
              open.pop_front();  // workers instead of queueing behind each other)
              open.push_back(r);
            }
            r->run();
          }
        });
    }
    ~Pool() {
      {
        std::lock_guard<std::mutex> g(mu);
        quit = true;
      }
      cv.notify_all();
      for (auto &t : workers) t.join();
    }
  };
  static Pool pool(nthreads);
  auto region = std::make_shared<Region>();
  region->body = &body;
  region->n = n;
  {
    std::lock_guard<std::mutex> g(pool.mu);
    pool.open.push_back(region);
  }
  pool.cv.notify_all();
  region->run();
  if (region->done.load() < n) {  // (workers still inside their last items: sleep, a spinning caller eats CPU quota)
    std::unique_lock<std::mutex> g(region->fin_mu);
    region->fin_cv.wait(g, [&] { return region->done.load() >= n; });
  }
  if (!region->error.empty()) throw region->error;
}

static std::vector<Hit> read_schedule(const std::string &bed_path, FILE *log) {  // src/align_main.cc:200-283, nbins=1
  std::ifstream fin(bed_path.c_str());
  if (!fin.is_open()) throw "BED file " + bed_path + " does not exist";
  // (the lines are parsed on the host threads: 40,000 lines take 26 ms on one)
  std::vector<std::string> text;
  std::string s;
  while (std::getline(fin, s)) text.push_back(std::move(s));
  std::vector<Hit> hits(text.size());
  parallel_for((int)text.size(), [&](int k) { hits[(size_t)k] = Hit::from_bed(text[(size_t)k]); });
  fprintf(log, "Read %d alignments in %s\n", (int)hits.size(), bed_path.c_str());
  fprintf(log, "Read total %d alignments\n", (int)hits.size());
  int max_complexity = 0;
  auto cx = [](const Hit &h) {
    return (int)std::sqrt(double(h.query_end - h.query_start) * double(h.ref_end - h.ref_start));
  };
  for (auto &h : hits) max_complexity = std::max(max_complexity, cx(h));
  // (a stable counting sort by bin, the records moved, not copied: a Hit owns two shared sequences and three strings)
  std::vector<int> bin(hits.size());
  std::vector<size_t> start(max_complexity / 1000 + 2, 0);
  for (size_t k = 0; k < hits.size(); k++) {
    bin[k] = cx(hits[k]) / 1000;
    start[(size_t)bin[k] + 1]++;
  }
  for (size_t b = 1; b < start.size(); b++) start[b] += start[b - 1];
  std::vector<Hit> order(hits.size());
  for (size_t k = 0; k < hits.size(); k++) order[start[(size_t)bin[k]]++] = std::move(hits[k]);
  return order;
}

GenerateStats generate_alignments(const std::string &ref_path, const std::string &bed_path, int kmer_size,
                                  const Params &p_in, DpProvider &dp0, FILE *out, FILE *log, int super_batch) {
  const auto t0 = std::chrono::steady_clock::now();
  Params p = p_in;
  p.kmer = kmer_size;
  set_alignment_scoring(p);
  GenerateStats st;
  struct Acc {  // wall seconds of the phases of one lane of the driver
    double dp_secs = 0, anchor_secs = 0, t_fetch = 0, t_adv = 0, t_longest = 0, t_sum = 0, t_collect = 0, t_out = 0;
    int rounds = 0;
  };
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point a) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
  };
  g_us_chain = 0;
  g_us_rest = 0;
  // SDF_DEBUG_TIMING: one line per phase of every super-batch, milliseconds since the stage clock started
  const bool dbg_tl = stage_settings().debug_timing;
  auto mark = [&](int base, const char *what) {
    if (dbg_tl) fprintf(stderr, "[stage %7.1f ms] batch@%d %s\n", since(t0) * 1e3, base, what);
  };
  std::vector<Hit> schedule = read_schedule(bed_path, log);
  mark(-1, "schedule read");
  FastaReference fr(ref_path);
  mark(-1, "fasta index open");
  fprintf(log, "Using k-mer size %d\n", kmer_size);
  const int total = (int)schedule.size();

  struct Item {
    Hit h;
    SeqView fa, fb;  // in the super-batch's pool (the provider's pinned staging, or the lane's own memory)
    std::unique_ptr<PairJob> job;
    std::vector<DpRequest> pending;
  };
  // Pairs per super-batch: every round of a super-batch is one device batch call (planning, launches, one
  // synchronisation), so few large super-batches beat many small ones; bounded by the sequence bytes held at once.
  // Lanes: super-batches are independent, so two or three of them are in flight, each on its own device context -- while
  // one waits for the device, the host threads work on the other.  A second context costs ~0.1 s to set up, so
  // small inputs stay on one lane; medium ones are cut into at least two super-batches per lane.
  std::vector<int> devices;
  int nlanes = stage_lane_count(total, &devices);
  super_batch = stage_super_batch(total, nlanes, super_batch);
  std::vector<std::pair<int, int>> batches;  // (first pair, pairs)
  int64_t max_batch_bytes = 0;
  for (int base = 0, n = 0; base < total; base += n) {
    int64_t bytes = 0;
    n = 0;
    while (base + n < total && n < super_batch && bytes < ((int64_t)1 << 30)) {
      const Hit &h = schedule[base + n];
      bytes += (int64_t)(h.query_end - h.query_start) + (h.ref_end - h.ref_start);
      ++n;
    }
    batches.push_back({base, n});
    max_batch_bytes = std::max(max_batch_bytes, bytes);
  }
  dp0.prepare((size_t)max_batch_bytes);
  auto zero_counters = [](DpProvider &d) {  // (a provider serves bucket after bucket of one process: the figures below are this run's)
    d.tasks = d.cells = 0;
    d.t_pack = d.t_call = d.t_unpack = 0;
  };
  zero_counters(dp0);

  std::mutex cleanup_mu;
  std::vector<std::thread> cleanup;
  struct JoinAll {  // (also on the way out with an exception)
    std::vector<std::thread> &v;
    ~JoinAll() {
      for (auto &t : v)
        if (t.joinable()) t.join();
    }
  } join_cleanup{cleanup};
  // One super-batch from the sequences to its formatted output lines (one string per pair, schedule order).
  auto do_batch = [&](int base, int n, DpProvider &dp, Acc &a, std::vector<std::string> &lines, std::vector<int> &nhits,
                      std::unique_ptr<char[]> &own_pool, size_t &own_pool_cap) {
    std::vector<Item> items(n);
    mark(base, "start");
    const auto tf = now();
    // src/align_main.cc:299-306.  The bases of the whole super-batch go into ONE pool -- the provider's pinned staging when it
    // has one: the anchors call uploads from there without another copy, and the DP rounds name ranges of it -- a slot per
    // sequence as long as the bytes its range spans in the file (line ends included: the bound known before the copy).
    std::vector<FastaReference::Span> span(2 * (size_t)n);
    std::vector<size_t> slot(2 * (size_t)n + 1, 0);
    for (int k = 0; k < n; k++) {
      Item &it = items[k];
      it.h = schedule[base + k];
      span[2 * k] = fr.locate(it.h.query->name, it.h.query_start, &it.h.query_end);
      span[2 * k + 1] = fr.locate(it.h.ref->name, it.h.ref_start, &it.h.ref_end);
      slot[2 * k + 1] = slot[2 * k] + span[2 * k].bytes;
      slot[2 * k + 2] = slot[2 * k + 1] + span[2 * k + 1].bytes;
    }
    char *pool = dp.pool_host(slot[2 * (size_t)n] + 1);
    const bool provider_pool = pool != nullptr;
    if (!pool) {
      if (own_pool_cap < slot[2 * (size_t)n] + 1) {
        own_pool_cap = slot[2 * (size_t)n] + 1 + slot[2 * (size_t)n] / 8;
        own_pool.reset(new char[own_pool_cap]);
      }
      pool = own_pool.get();
    }
    parallel_for(n, [&](int k) {
      Item &it = items[k];
      char *qa = pool + slot[2 * k], *ra = pool + slot[2 * k + 1];
      it.fa = SeqView(qa, FastaReference::extract(span[2 * k], qa));
      it.fb = SeqView(ra, FastaReference::extract(span[2 * k + 1], ra));
      if (it.h.ref->is_rc) rc_inplace(ra, it.fb.size());
      it.job.reset(new PairJob(it.fa, it.fb, it.h, p));
    });
    a.t_fetch += since(tf);
    dp.pool_ready(slot[2 * (size_t)n]);  // (the whole pool to the device, asynchronously)
    mark(base, "sequences fetched");
    // Seed anchors on the device, when the provider offers it -- in PARTS of about equal bytes (two from 16 MB of sequences,
    // four from 64 MB): while the device finds the anchors of part i + 1, the host threads chain part i (a pair's first
    // advance(): src/chain.cc:203-258 up to the requests of its round-A stitch), which is the longest host phase of a
    // super-batch (20 of 85 ms in the chr1-sized run).  Every part's anchors are appended behind those of the parts before
    // it in the provider's staging (anchors_more), which stay valid until their jobs have taken their copies.
    std::vector<DpProvider::AnchorBatch> seeds;  // (live until the jobs have taken their copies: their first advance)
    std::vector<int64_t> q_base((size_t)n, 0), r_base((size_t)n, 0);
    bool resident = false;
    std::vector<char> advanced((size_t)n, 0);  // pairs whose first advance() has run (their requests wait in `pending`)
    std::atomic<long long> pre_us(0);
    {
      std::vector<DpProvider::AnchorJob> aj(n);
      for (int k = 0; k < n; k++) {
        const Hit &h = items[k].h;
        aj[k] = {items[k].fa, items[k].fb, h.query->name == h.ref->name && h.query->is_rc == h.ref->is_rc,
                 h.ref_start - h.query_start};
      }
      // (only where the sequences lie in the PROVIDER's pool: a provider that copies them replaces its pool with every call)
      const size_t total = slot[2 * (size_t)n];
      int parts = 1;
      if (provider_pool && n >= 64 && total >= ((size_t)16 << 20)) parts = 2;  // (small super-batches: one call)
      if (provider_pool && n >= 256 && total >= ((size_t)64 << 20)) parts = 4;
      if (stage_settings().anchor_parts > 0 && provider_pool) parts = std::min(stage_settings().anchor_parts, std::max(n / 16, 1));
      std::vector<int> cut((size_t)parts + 1, n);  // part i: pairs [cut[i], cut[i + 1])
      cut[0] = 0;
      for (int i = 1, k = 0; i < parts; i++) {
        while (k < n && slot[2 * (size_t)k] < total / (size_t)parts * (size_t)i) ++k;
        cut[(size_t)i] = k;
      }
      for (int i = 0; i < parts; i++)
        if (cut[(size_t)i + 1] - cut[(size_t)i] < 16) {  // (a part of a few pairs: one call for everything)
          parts = 1;
          cut.assign(2, n);
          cut[0] = 0;
          break;
        }
      seeds.resize((size_t)parts);
      auto first_advance = [&](int lo, int hi, const DpProvider::AnchorBatch &sd) {
        parallel_for(hi - lo, [&](int i) {
          const int k = lo + i;
          Item &it = items[k];
          const auto tj = std::chrono::steady_clock::now();
          it.job->set_anchors(sd.data() + sd.off[i], (size_t)(sd.off[i + 1] - sd.off[i]));
          it.pending = it.job->advance(std::vector<Cigar>());
          advanced[(size_t)k] = 1;
          pre_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tj).count();
        });
      };
      auto part_jobs = [&](int i) {
        return std::vector<DpProvider::AnchorJob>(aj.begin() + cut[(size_t)i], aj.begin() + cut[(size_t)i + 1]);
      };
      const auto ta = now();
      if (dp.anchors(part_jobs(0), p.kmer, seeds[0])) {
        resident = seeds[0].resident;
        auto take_bases = [&](int i) {
          for (int k = cut[(size_t)i]; k < cut[(size_t)i + 1]; k++) {
            q_base[k] = resident ? seeds[(size_t)i].q_base[k - cut[(size_t)i]] : 0;
            r_base[k] = resident ? seeds[(size_t)i].r_base[k - cut[(size_t)i]] : 0;
          }
        };
        take_bases(0);
        if (parts == 1) {
          for (int k = 0; k < n; k++)
            items[k].job->set_anchors(seeds[0].data() + seeds[0].off[k], (size_t)(seeds[0].off[k + 1] - seeds[0].off[k]));
        } else {
          size_t keep = (size_t)seeds[0].off[cut[1]];  // anchors in the staging so far
          bool have = true;                            // part i's anchors are there
          for (int i = 0; i < parts; i++) {
            const bool more = i + 1 < parts;
            bool ok_next = false;
            std::string err_next;
            std::vector<DpProvider::AnchorJob> next_jobs;
            std::thread next;
            if (more) {
              next_jobs = part_jobs(i + 1);
              next = std::thread([&] {
                try {
                  ok_next = dp.anchors_more(next_jobs, p.kmer, keep, seeds[(size_t)i + 1]);
                } catch (std::string &e) {
                  err_next = e.empty() ? std::string("error") : e;
                }
              });
            }
            struct Join {
              std::thread &t;
              ~Join() {
                if (t.joinable()) t.join();
              }
            } join_next{next};
            if (i == 0) mark(base, "anchors of the first part done");
            // (a part without anchors from the device: its jobs find theirs on the host, in their first advance below)
            if (have) first_advance(cut[(size_t)i], cut[(size_t)i + 1], seeds[(size_t)i]);
            if (!more) break;
            next.join();
            if (!err_next.empty()) throw err_next;
            // (no room behind the anchors of the parts before, or a part the device does not cover: the ordinary call, now
            // that the earlier parts' jobs have taken their copies -- it starts the staging over)
            bool fresh = false;
            if (!ok_next) ok_next = fresh = dp.anchors(next_jobs, p.kmer, seeds[(size_t)i + 1]);
            have = ok_next;
            if (!ok_next) resident = false;  // (its pairs have no place in the resident pool the requests could name)
            if (ok_next && seeds[(size_t)i + 1].resident != resident) resident = false;  // (the parts disagree about where the
                                                                                        // sequences are: the pointer form for everybody)
            if (ok_next && resident) take_bases(i + 1);
            if (ok_next) keep = (fresh ? 0 : keep) + (size_t)seeds[(size_t)i + 1].off[cut[(size_t)i + 2] - cut[(size_t)i + 1]];
          }
          mark(base, "anchors done");
        }
        a.anchor_secs += since(ta);
      }
    }
    mark(base, "anchors done, parts chained");
    // rounds: every unfinished job advances; all their DP requests go to the GPU as one batch.
    // Results of the previous round: Cigars per pair (provider without a raw form), or the raw device words
    // and each pair's first request in them -- then the pair's own thread builds (and later frees) its Cigars
    std::vector<std::vector<Cigar>> results(n);
    DpProvider::Raw raw;
    bool have_raw = false;
    std::vector<size_t> first_req(n, 0), n_req(n, 0);
    for (;;) {
      std::vector<DpRequest> batch;
      std::vector<DpProvider::ResidentReq> rbatch;
      std::vector<std::pair<int, size_t>> owners;
      bool any = false;
      const auto tadv = now();
      std::atomic<long long> longest_us(0), sum_us(0);
      parallel_for(n, [&](int k) {
        Item &it = items[k];
        if (advanced[(size_t)k]) {  // (its first advance ran next to the second half's anchors: the requests are waiting)
          advanced[(size_t)k] = 0;
          return;
        }
        it.pending.clear();
        if (it.job->done()) return;
        const auto tj = std::chrono::steady_clock::now();
        if (have_raw) {
          std::vector<Cigar> mine(n_req[k]);
          for (size_t r = 0; r < n_req[k]; r++) mine[r] = raw.cigar(first_req[k] + r);
          it.pending = it.job->advance(mine);
        } else {
          it.pending = it.job->advance(results[k]);
          results[k].clear();
        }
        const long long us =
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tj).count();
        sum_us += us;
        long long cur = longest_us.load();
        while (us > cur && !longest_us.compare_exchange_weak(cur, us)) {
        }
      });
      a.t_adv += since(tadv);
      a.t_longest += longest_us.load() / 1e6;
      a.t_sum += (sum_us.load() + pre_us.exchange(0)) / 1e6;
      const auto tc = now();
      std::vector<size_t> first(n + 1, 0);
      for (int k = 0; k < n; k++) {
        first[k + 1] = first[k] + items[k].pending.size();
        if (!items[k].pending.empty()) {
          owners.push_back({k, items[k].pending.size()});
          any = true;
        }
      }
      // (requests are four words each; with the pairs' characters on the device they become offsets of that pool, on the
      // host threads: 708,600 of them in the first round of the chr1-sized run)
      if (resident) rbatch.resize(first[n]);
      else batch.resize(first[n]);
      std::atomic<bool> outside(false);
      parallel_for(n, [&](int k) {
        Item &it = items[k];
        size_t at = first[k];
        if (resident) {
          const char *qa = it.fa.data(), *ra = it.fb.data();
          for (const DpRequest &r : it.pending) {
            if (r.q < qa || r.q + r.qlen > qa + it.fa.size() || r.t < ra || r.t + r.tlen > ra + it.fb.size()) outside.store(true);
            rbatch[at++] = {q_base[k] + (r.q - qa), r_base[k] + (r.t - ra), r.qlen, r.tlen};
          }
        } else {
          for (const DpRequest &r : it.pending) batch[at++] = r;
        }
        it.pending.clear();
      });
      if (outside.load()) throw std::string("internal: a DP request outside its pair's sequences");
      a.t_collect += since(tc);
      if (!any) break;
      a.rounds++;
      mark(base, "jobs advanced, requests collected");
      const auto td = now();
      std::vector<Cigar> got;
      have_raw = resident ? dp.run_resident(rbatch, p, raw) : dp.run_raw(batch, p, raw);
      if (resident && !have_raw) throw std::string("internal: the provider lost its resident sequences");
      if (!have_raw) got = dp.run(batch, p);
      a.dp_secs += since(td);
      mark(base, "DP round done");
      const auto tc2 = now();
      size_t cur = 0;
      std::fill(n_req.begin(), n_req.end(), 0);
      for (auto &o : owners) {
        if (have_raw) {
          first_req[o.first] = cur;
          n_req[o.first] = o.second;
        } else {
          results[o.first].assign(std::make_move_iterator(got.begin() + cur),
                                  std::make_move_iterator(got.begin() + cur + o.second));
        }
        cur += o.second;
      }
      a.t_collect += since(tc2);
    }
    const auto tout = now();
    lines.assign(n, std::string());  // formatted on the host threads, written in schedule order
    nhits.assign(n, 0);
    parallel_for(n, [&](int k) {  // src/align_main.cc:314-331
      Item &it = items[k];
      const std::string tail = "\t" + it.h.to_bed(false) + "\n";
      for (auto &hh : it.job->hits()) {
        hh.query_start += it.h.query_start;
        hh.query_end += it.h.query_start;
        if (it.h.ref->is_rc) {
          std::swap(hh.ref_start, hh.ref_end);
          hh.ref_start = it.h.ref_end - hh.ref_start;
          hh.ref_end = it.h.ref_end - hh.ref_end;
          hh.ref->is_rc = true;
        } else {
          hh.ref_start += it.h.ref_start;
          hh.ref_end += it.h.ref_start;
        }
        hh.query->name = it.h.query->name;
        hh.ref->name = it.h.ref->name;
        nhits[k]++;
        lines[k] += hh.to_bed(false);
        lines[k] += tail;
      }
    });
    a.t_out += since(tout);
    mark(base, "output formatted");
    // (the pairs' jobs hold hundreds of megabytes in small pieces: freeing them took a lane 15-50 ms between two
    // super-batches -- a thread of its own does it while the lane fetches the next one's sequences)
    auto *gone = new std::vector<Item>(std::move(items));
    std::lock_guard<std::mutex> g(cleanup_mu);
    cleanup.emplace_back([gone] { delete gone; });
  };
  auto write_batch = [&](int base, int n, const std::vector<std::string> &lines, const std::vector<int> &nhits) {
    for (int k = 0; k < n; k++) {
      st.lines++;
      st.total_written += nhits[k];
      if (!lines[k].empty()) fwrite(lines[k].data(), 1, lines[k].size(), out);
    }
    fprintf(log, "\r Processing %d out of %d (%.1f%%)", std::min(base + n, total), total,
            100.0 * std::min(base + n, total) / std::max(total, 1));
  };

  // Output is written in schedule order whatever lane produced it.  Every lane but the first creates its own device
  // context (~0.1 s) on its own thread while the first one is already working; the super-batches are handed out as the
  // lanes ask for them, so a lane that starts late, or cannot get a context at all, just takes fewer.
  nlanes = std::min<int>(nlanes, (int)batches.size());
  std::vector<std::unique_ptr<DpProvider>> extra((size_t)std::max(nlanes, 1));
  std::vector<DpProvider *> prov((size_t)std::max(nlanes, 1), nullptr);
  prov[0] = &dp0;
  std::vector<Acc> acc((size_t)std::max(nlanes, 1));
  if (nlanes <= 1) {
    std::vector<std::string> lines;
    std::vector<int> nhits;
    std::unique_ptr<char[]> own_pool;  // (the lane's sequence pool when its provider has none: kept from batch to batch)
    size_t own_pool_cap = 0;
    for (auto &b : batches) {
      do_batch(b.first, b.second, dp0, acc[0], lines, nhits, own_pool, own_pool_cap);
      write_batch(b.first, b.second, lines, nhits);
    }
  } else {
    // The super-batches with the most sequence first (the schedule is sorted by size: its last super-batch is the
    // heaviest, and taken last it ran on alone while the other lanes had nothing left); a finished super-batch keeps its
    // lines until every earlier one of the schedule has been written.
    std::vector<size_t> turn(batches.size());
    {
      std::vector<int64_t> weight(batches.size(), 0);
      for (size_t b = 0; b < batches.size(); b++)
        for (int k = 0; k < batches[b].second; k++) {
          const Hit &h = schedule[(size_t)(batches[b].first + k)];
          weight[b] += (int64_t)(h.query_end - h.query_start) + (h.ref_end - h.ref_start);
        }
      for (size_t b = 0; b < turn.size(); b++) turn[b] = b;
      std::stable_sort(turn.begin(), turn.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
    }
    std::vector<std::vector<std::string>> done_lines(batches.size());
    std::vector<std::vector<int>> done_nhits(batches.size());
    std::vector<char> is_done(batches.size(), 0);
    std::mutex mu;
    size_t next_write = 0;
    std::atomic<size_t> next_batch(0);
    std::string failure;
    bool failed = false;
    std::vector<std::thread> lanes;
    for (int l = 0; l < nlanes; l++)
      lanes.emplace_back([&, l] {
        if (l > 0) {
          try {
            extra[(size_t)l] = dp0.clone(devices.empty() ? -1 : devices[(size_t)l % devices.size()]);
          } catch (std::string &) {  // no room for another device context: one lane fewer
          }
          prov[(size_t)l] = extra[(size_t)l].get();
          mark(-2 - l, "lane's device context ready");
          if (!prov[(size_t)l]) return;
          zero_counters(*prov[(size_t)l]);
          prov[(size_t)l]->prepare((size_t)max_batch_bytes);
        }
        std::vector<std::string> lines;
        std::vector<int> nhits;
        std::unique_ptr<char[]> own_pool;
        size_t own_pool_cap = 0;
        for (;;) {
          const size_t ti = next_batch.fetch_add(1);
          if (ti >= batches.size()) return;
          const size_t bi = turn[ti];
          try {
            do_batch(batches[bi].first, batches[bi].second, *prov[(size_t)l], acc[(size_t)l], lines, nhits, own_pool, own_pool_cap);
          } catch (std::string &e) {
            std::lock_guard<std::mutex> g(mu);
            if (!failed) failure = e.empty() ? std::string("error") : e;
            failed = true;
            return;
          }
          std::lock_guard<std::mutex> g(mu);
          if (failed) return;
          done_lines[bi].swap(lines);
          done_nhits[bi].swap(nhits);
          is_done[bi] = 1;
          while (next_write < batches.size() && is_done[next_write]) {  // (whoever completes the next one in line writes)
            write_batch(batches[next_write].first, batches[next_write].second, done_lines[next_write], done_nhits[next_write]);
            std::vector<std::string>().swap(done_lines[next_write]);
            ++next_write;
          }
        }
      });
    for (auto &t : lanes) t.join();
    if (failed) throw failure;
  }
  const double secs = since(t0);  // (the output is complete; what is left is giving memory back)
  for (auto &t : cleanup) t.join();
  cleanup.clear();
  Acc a;
  for (int l = 0; l < nlanes; l++) {
    a.dp_secs += acc[l].dp_secs;
    a.anchor_secs += acc[l].anchor_secs;
    a.t_fetch += acc[l].t_fetch;
    a.t_adv += acc[l].t_adv;
    a.t_longest += acc[l].t_longest;
    a.t_sum += acc[l].t_sum;
    a.t_collect += acc[l].t_collect;
    a.t_out += acc[l].t_out;
    a.rounds += acc[l].rounds;
  }
  st.rounds = a.rounds;
  double t_pack = 0, t_call = 0, t_unpack = 0;
  for (DpProvider *d : prov) {
    if (!d) continue;
    st.dp_tasks += d->tasks;
    st.dp_cells += d->cells;
    t_pack += d->t_pack;
    t_call += d->t_call;
    t_unpack += d->t_unpack;
  }
  fprintf(log, "\nFinished BED %s in %.2fs (%d lines, generated %d hits)\n", bed_path.c_str(), secs, st.lines,
          st.total_written);
  fprintf(log, "  [%d lane(s); host CPU: anchors+chaining %.2fs, stitching+refinement %.2fs (summed over threads); device "
               "anchors %.2fs; DP provider %.2fs in %d rounds, %lld tasks, %.3g cells]\n",
          nlanes, g_us_chain.load() / 1e6, g_us_rest.load() / 1e6, a.anchor_secs, a.dp_secs, st.rounds,
          (long long)st.dp_tasks, (double)st.dp_cells);
  fprintf(log, "  [driver (summed over lanes): sequence fetch %.2fs, job rounds on host threads %.2fs (longest single jobs "
               "%.2fs, all jobs %.2fs thread time), request collection %.2fs, output %.2fs; DP provider: request "
               "packing %.2fs, device call %.2fs, CIGAR unpacking %.2fs]\n",
          a.t_fetch, a.t_adv, a.t_longest, a.t_sum, a.t_collect, a.t_out, t_pack, t_call, t_unpack);
  // (the extra lanes' providers go back to the one they came from: the next bucket of this process takes them again, and the
  // first provider gives all their device contexts back side by side when it goes)
  for (auto &e : extra)
    if (e) dp0.give_back(std::move(e));
  return st;
}

// ---- several buckets, one process --------------------------------------------------------------------------------
std::vector<std::string> expand_buckets(const std::vector<std::string> &beds) {
  std::vector<std::string> out;
  for (const std::string &b : beds) {
    struct stat sb;
    if (stat(b.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode)) {
      // the bucket files `sedef align bucket` wrote there (src/align_main.cc:178: "bucket_{:04d}"), not the outputs next to them
      std::vector<std::string> names;
      if (DIR *d = opendir(b.c_str())) {
        while (struct dirent *e = readdir(d)) {
          const std::string n = e->d_name;
          if (n.size() == 11 && n.compare(0, 7, "bucket_") == 0 && n.find_first_not_of("0123456789", 7) == std::string::npos)
            names.push_back(n);
        }
        closedir(d);
      }
      std::sort(names.begin(), names.end());
      for (auto &n : names) out.push_back(b + (b.empty() || b.back() == '/' ? "" : "/") + n);
    } else {
      out.push_back(b);
    }
  }
  return out;
}

StageHint stage_hint_many(const std::vector<std::string> &beds, int super_batch) {
  StageHint all;
  all.pairs = 0;
  for (const std::string &b : beds) {
    const StageHint h = stage_hint(b, super_batch);
    all.pairs += h.pairs;
    if (h.lanes > all.lanes) {
      all.lanes = h.lanes;
      all.devices = h.devices;
    }
    all.super_batch = std::max(all.super_batch, h.super_batch);
    all.max_batch_bytes = std::max(all.max_batch_bytes, h.max_batch_bytes);
  }
  // (one-lane buckets run SDF_BUCKET_LANES at a time, generate_many: a provider each)
  if (all.lanes <= 1 && beds.size() >= 2 && stage_settings().bucket_lanes > 1)
    all.lanes = (int)std::min<size_t>((size_t)stage_settings().bucket_lanes, beds.size());
  return all;
}

std::vector<GenerateStats> generate_many(const std::string &ref_path, const std::vector<std::string> &beds, int kmer_size,
                                         const Params &p, DpProvider &dp, const std::string &out_suffix,
                                         const std::string &log_dir, FILE *log, int super_batch) {
  // One bucket's phases use the host and the device in turn (sequence fetch, anchors, ~65 ms of chaining on all threads, DP
  // rounds, output): with several one-lane buckets to do, SDF_BUCKET_LANES of them (default 2) are in flight, each on a
  // provider -- a device context -- of its own, taking buckets as they go.  A bucket's output file and log are its own,
  // its lines are what a process of its own writes; the order of the "Finished" lines on the shared log is the order of
  // completion.  (Buckets large enough for several lanes of super-batches run one after the other: their lanes fill the gaps.)
  size_t conc = 1;
  if (beds.size() >= 2 && stage_settings().bucket_lanes > 1) {
    bool one_lane = true;
    for (const std::string &bed : beds) one_lane = one_lane && stage_hint(bed, super_batch).lanes <= 1;
    if (one_lane) conc = std::min<size_t>((size_t)stage_settings().bucket_lanes, beds.size());
  }
  if (conc > 1) {
    Params pk = p;
    pk.kmer = kmer_size;
    set_alignment_scoring(pk);  // (before the threads: generate_alignments finds it in place)
    std::vector<GenerateStats> all(beds.size());
    std::vector<std::unique_ptr<DpProvider>> extra(conc);
    std::atomic<size_t> next(0);
    std::mutex mu;
    std::string failure;
    bool failed = false;
    auto run_lane = [&](size_t l) {
      DpProvider *d = &dp;
      if (l > 0) {
        try {
          extra[l] = dp.clone(-1);
        } catch (std::string &) {  // (no room for another device context: one bucket lane fewer)
        }
        d = extra[l].get();
        if (!d) return;
      }
      for (;;) {
        const size_t bi = next.fetch_add(1);
        if (bi >= beds.size()) return;
        {
          std::lock_guard<std::mutex> g(mu);
          if (failed) return;
        }
        const std::string &bed = beds[bi];
        const std::string out_path = bed + out_suffix;
        FILE *out = fopen(out_path.c_str(), "w");
        FILE *blog = nullptr;
        std::string err;
        if (!out) err = std::string("Cannot open file ") + out_path + " for writing";
        if (err.empty() && !log_dir.empty()) {
          const size_t slash = bed.find_last_of('/');
          const std::string lp = log_dir + "/" + (slash == std::string::npos ? bed : bed.substr(slash + 1)) + ".log";
          blog = fopen(lp.c_str(), "w");
          if (!blog) err = std::string("Cannot open file ") + lp + " for writing";
        }
        // (without --log-dir the buckets' own progress lines would interleave on the shared log: they go to a buffer that is
        // written in one piece when the bucket is done)
        char *membuf = nullptr;
        size_t memlen = 0;
        FILE *mlog = (!blog && err.empty()) ? open_memstream(&membuf, &memlen) : nullptr;
        if (err.empty()) {
          try {
            all[bi] = generate_alignments(ref_path, bed, kmer_size, p, *d, out, blog ? blog : mlog ? mlog : log, super_batch);
          } catch (std::string &e) {
            err = e.empty() ? std::string("error") : e;
          } catch (std::exception &e) {
            err = e.what();
          }
        }
        if (out) fclose(out);
        if (blog) fclose(blog);
        if (mlog) fclose(mlog);
        std::lock_guard<std::mutex> g(mu);
        if (membuf) {
          if (memlen) fwrite(membuf, 1, memlen, log);
          free(membuf);
        }
        if (!err.empty()) {
          if (!failed) failure = err;
          failed = true;
          return;
        }
        if (blog) fprintf(log, "Finished BED %s (%d lines, generated %d hits)\n", bed.c_str(), all[bi].lines, all[bi].total_written);
      }
    };
    std::vector<std::thread> lanes;
    for (size_t l = 1; l < conc; l++) lanes.emplace_back(run_lane, l);
    run_lane(0);
    for (auto &t : lanes) t.join();
    for (auto &e : extra)
      if (e) dp.give_back(std::move(e));
    if (failed) throw failure;
    return all;
  }
  std::vector<GenerateStats> all;
  for (const std::string &bed : beds) {
    const std::string out_path = bed + out_suffix;
    FILE *out = fopen(out_path.c_str(), "w");
    if (!out) throw std::string("Cannot open file ") + out_path + " for writing";
    FILE *blog = nullptr;
    if (!log_dir.empty()) {
      const size_t slash = bed.find_last_of('/');
      const std::string lp = log_dir + "/" + (slash == std::string::npos ? bed : bed.substr(slash + 1)) + ".log";
      blog = fopen(lp.c_str(), "w");
      if (!blog) {
        fclose(out);
        throw std::string("Cannot open file ") + lp + " for writing";
      }
    }
    try {
      all.push_back(generate_alignments(ref_path, bed, kmer_size, p, dp, out, blog ? blog : log, super_batch));
    } catch (...) {
      fclose(out);
      if (blog) fclose(blog);
      throw;
    }
    fclose(out);
    if (blog) {
      fclose(blog);
      fprintf(log, "Finished BED %s (%d lines, generated %d hits)\n", bed.c_str(), all.back().lines, all.back().total_written);
    }
  }
  return all;
}

}  // namespace sdfh
