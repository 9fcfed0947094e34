// sdf_multi: the north star's multi-GPU flow as host C++ above the C ABI -- ONE batch of DP tasks sharded over the GPUs of a
// node by cells, every GPU aligns its shard (sdf_extz2_batch_device), and an RCCL all-gatherv over xGMI
// (sdf_allgatherv_results) gives every GPU every shard's result records and CIGAR words; the host then reads everything from
// the first device and checks it against what each device computed.
//
// The reference has no counterpart: `sedef align` runs one single-threaded process per bucket file (sedef.sh:187-190) and the
// processes meet in files (sedef.sh:218-221).  One process, one thread per GPU, ncclCommInitAll through sdf_comm_create_all.
//
//   sdf_multi [--devices 0,1,...] [--tasks 100000] [--qlen 1000] [--band 128] [--steps 5] [--warmup 2]
// prints one JSON line: whole-job Gcell/s (max over the devices' clocks), the shard balance and the union check.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <queue>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/sedef_hip.h"

namespace {

struct Barrier {  // (std::barrier is C++20)
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0, gen = 0;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int g = gen;
    if (++waiting == n) {
      waiting = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

struct Batch {
  std::vector<uint8_t> codes;
  std::vector<int64_t> q_off, t_off;
  std::vector<int32_t> qlen, tlen;
  std::vector<int64_t> cells;
};

// query = qlen uniform ACGT, target = query with 6 % substitution draws, 2 % deletions, 2 % insertions
Batch synth(size_t n, int qlen, int w, unsigned seed) {
  Batch b;
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  b.codes.reserve(n * (size_t)(2 * qlen + 64));
  for (size_t k = 0; k < n; ++k) {
    b.q_off.push_back((int64_t)b.codes.size());
    std::vector<uint8_t> q(qlen);
    for (auto &c : q) c = (uint8_t)(rng() & 3);
    b.codes.insert(b.codes.end(), q.begin(), q.end());
    b.t_off.push_back((int64_t)b.codes.size());
    int tl = 0;
    for (int i = 0; i < qlen; ++i) {
      const double x = u(rng);
      if (x < 0.06) {
        b.codes.push_back((uint8_t)(rng() & 3)), ++tl;
      } else if (x < 0.08) {
      } else if (x < 0.10) {
        b.codes.push_back(q[i]), b.codes.push_back((uint8_t)(rng() & 3)), tl += 2;
      } else {
        b.codes.push_back(q[i]), ++tl;
      }
    }
    if (tl == 0) b.codes.push_back(q[0]), tl = 1;
    b.qlen.push_back(qlen);
    b.tlen.push_back(tl);
    b.cells.push_back(sdf_band_cells(qlen, tl, w));
  }
  return b;
}

// longest-processing-time-first by cells: shard[r] = the task indices of device r, ascending
std::vector<std::vector<size_t>> shard_by_cells(const std::vector<int64_t> &cells, int world) {
  std::vector<size_t> order(cells.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cells[a] > cells[b]; });
  typedef std::pair<int64_t, int> Load;
  std::priority_queue<Load, std::vector<Load>, std::greater<Load>> pq;
  for (int r = 0; r < world; ++r) pq.push({0, r});
  std::vector<std::vector<size_t>> shard(world);
  for (size_t i : order) {
    Load l = pq.top();
    pq.pop();
    shard[l.second].push_back(i);
    pq.push({l.first + cells[i], l.second});
  }
  for (auto &s : shard) std::sort(s.begin(), s.end());
  return shard;
}

uint64_t checksum(const sdf_result &r, const uint32_t *cig) {  // score, n_cigar and the CIGAR words of one task
  uint64_t h = (uint64_t)(uint32_t)r.score * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)r.n_cigar;
  for (int j = 0; j < r.n_cigar; ++j) h = (h ^ cig[r.cigar_off + j]) * 0x100000001B3ull;
  return h;
}

#define CHECK_HIP(call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "sdf_multi: %s: %s\n", #call, hipGetErrorString(e_));                  \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

}  // namespace

int main(int argc, char **argv) {
  std::vector<int> devices;
  size_t n = 100000;
  int qlen = 1000, w = 128, steps = 5, warmup = 2;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--devices") {
      const std::string v = val();
      for (size_t p = 0; p < v.size();) {
        const size_t e = v.find(',', p);
        devices.push_back(atoi(v.substr(p, e == std::string::npos ? e : e - p).c_str()));
        if (e == std::string::npos) break;
        p = e + 1;
      }
    } else if (a == "--tasks") n = (size_t)atoll(val());
    else if (a == "--qlen") qlen = atoi(val());
    else if (a == "--band") w = atoi(val());
    else if (a == "--steps") steps = atoi(val());
    else if (a == "--warmup") warmup = atoi(val());
    else {
      fprintf(stderr, "usage: sdf_multi [--devices 0,1,...] [--tasks N] [--qlen L] [--band W] [--steps K] [--warmup K]\n");
      return 2;
    }
  }
  if (devices.empty())
    for (int d = 0; d < std::max(1, sdf_device_count()); ++d) devices.push_back(d);
  const int world = (int)devices.size();
  const Batch b = synth(n, qlen, w, 42);
  const auto shards = shard_by_cells(b.cells, world);
  std::vector<sdf_comm *> comms(world, nullptr);
  if (sdf_comm_create_all(devices.data(), world, comms.data()) != SDF_OK) {
    fprintf(stderr, "sdf_multi: %s\n", sdf_comm_last_error(nullptr));
    return 1;
  }
  sdf_scoring sc;
  memset(&sc, 0, sizeof(sc));
  sc.m = 5;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) sc.mat[i * 5 + j] = i == j ? 5 : -4;
  sc.gapo = 40, sc.gape = 1;
  Barrier bar(world);
  std::vector<double> secs(world, 0.0);
  std::vector<int64_t> shard_cells(world, 0);
  std::atomic<int> failures{0};
  std::vector<std::string> notes(world);
  auto rank_main = [&](int r) {
    CHECK_HIP(hipSetDevice(devices[r]));
    sdf_ctx *ctx = sdf_create(devices[r], (size_t)32 << 30);
    if (!ctx) {
      fprintf(stderr, "sdf_multi: device %d: %s\n", devices[r], sdf_last_error(nullptr));
      exit(1);
    }
    const std::vector<size_t> &mine = shards[r];
    const size_t m = mine.size();
    // this rank's shard, packed (2 bits a base + N mask) and resident in HBM
    std::vector<int64_t> q_off(m), t_off(m), q_word(m), t_word(m);
    std::vector<int32_t> ql(m), tl(m);
    size_t words = 0, cig_cap = 16;
    for (size_t k = 0; k < m; ++k) {
      q_off[k] = b.q_off[mine[k]], t_off[k] = b.t_off[mine[k]], ql[k] = b.qlen[mine[k]], tl[k] = b.tlen[mine[k]];
      words += sdf_packed_words(ql[k]) + sdf_packed_words(tl[k]);
      cig_cap += (size_t)ql[k] + tl[k] + 2;
      shard_cells[r] += b.cells[mine[k]];
    }
    std::vector<uint32_t> packed(words + 1);
    sdf_pack_tasks(b.codes.data(), q_off.data(), ql.data(), t_off.data(), tl.data(), m, packed.data(), q_word.data(), t_word.data());
    std::vector<sdf_task> tasks(m);
    for (size_t k = 0; k < m; ++k) {
      memset(&tasks[k], 0, sizeof(sdf_task));
      tasks[k].q_off = q_word[k], tasks[k].t_off = t_word[k], tasks[k].qlen = ql[k], tasks[k].tlen = tl[k];
      tasks[k].w = w, tasks[k].zdrop = -1;
    }
    uint32_t *d_pool, *d_cig, *d_all_cig;
    sdf_result *d_out, *d_all_out;
    size_t all_cig_cap = 16;
    for (size_t k = 0; k < n; ++k) all_cig_cap += (size_t)b.qlen[k] + b.tlen[k] + 2;
    // (qlen + tlen + 2 words per task bound any CIGAR, as the stage sizes its pools: no guess per band)
    CHECK_HIP(hipMalloc((void **)&d_pool, (words + 1) * 4));
    CHECK_HIP(hipMalloc((void **)&d_cig, cig_cap * 4));
    CHECK_HIP(hipMalloc((void **)&d_out, (m + 1) * sizeof(sdf_result)));
    CHECK_HIP(hipMalloc((void **)&d_all_out, (n + 1) * sizeof(sdf_result)));
    CHECK_HIP(hipMalloc((void **)&d_all_cig, all_cig_cap * 4));
    CHECK_HIP(hipMemcpy(d_pool, packed.data(), words * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> counts(2 * world);
    size_t used = 0;
    auto step = [&]() {
      if (m && sdf_extz2_batch_device(ctx, &sc, tasks.data(), m, d_pool, SDF_WANT_CIGAR | SDF_WANT_SCORE, d_out, d_cig, cig_cap, &used,
                                      nullptr) != SDF_OK) {
        fprintf(stderr, "sdf_multi: device %d: %s\n", devices[r], sdf_last_error(ctx));
        exit(1);
      }
      if (sdf_allgatherv_results(comms[r], d_out, m, d_cig, used, d_all_out, n + 1, d_all_cig, all_cig_cap, counts.data(), nullptr) !=
          SDF_OK) {
        fprintf(stderr, "sdf_multi: device %d: all-gatherv: %s\n", devices[r], sdf_comm_last_error(comms[r]));
        exit(1);
      }
    };
    for (int s = 0; s < warmup; ++s) step();
    CHECK_HIP(hipDeviceSynchronize());
    bar.wait();
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) step();
    CHECK_HIP(hipDeviceSynchronize());
    bar.wait();
    secs[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // ---- every device holds every shard's results: its own part is what it computed, and (first device) the union is the
    // whole batch, one record per task ----
    std::vector<sdf_result> own(m), all(n);
    std::vector<uint32_t> own_cig(used);
    uint64_t rec_total = 0, cig_total = 0, rec_before = 0, cig_before = 0;
    for (int q = 0; q < world; ++q) {
      if (q < r) rec_before += counts[2 * q], cig_before += counts[2 * q + 1];
      rec_total += counts[2 * q], cig_total += counts[2 * q + 1];
    }
    std::vector<uint32_t> all_cig(cig_total);
    CHECK_HIP(hipMemcpy(own.data(), d_out, m * sizeof(sdf_result), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(own_cig.data(), d_cig, used * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(all.data(), d_all_out, rec_total * sizeof(sdf_result), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(all_cig.data(), d_all_cig, cig_total * 4, hipMemcpyDeviceToHost));
    bool ok = rec_total == n && counts[2 * r] == m && counts[2 * r + 1] == used;
    for (size_t k = 0; ok && k < m; ++k)
      ok = checksum(own[k], own_cig.data()) == checksum(all[rec_before + k], all_cig.data() + cig_before);
    if (!ok) {
      ++failures;
      notes[r] = "gathered part differs from what the device computed";
    }
    if (r == 0 && ok) {  // the other ranks' parts: complete CIGARs that consume both sequences
      uint64_t ro = 0, co = 0;
      for (int q = 0; q < world && ok; ++q) {
        for (size_t k = 0; k < counts[2 * q] && ok; ++k) {
          const sdf_result &x = all[ro + k];
          const size_t task = shards[q][k];
          int64_t cq = 0, ct = 0;
          for (int j = 0; j < x.n_cigar; ++j) {
            const uint32_t wd = all_cig[co + x.cigar_off + j];
            if ((wd & 15) != 2) cq += wd >> 4;
            if ((wd & 15) != 1) ct += wd >> 4;
          }
          ok = x.n_cigar > 0 && cq == b.qlen[task] && ct == b.tlen[task];
        }
        ro += counts[2 * q], co += counts[2 * q + 1];
      }
      if (!ok) {
        ++failures;
        notes[0] = "a gathered CIGAR does not consume its task's sequences";
      }
    }
    (void)hipFree(d_pool), (void)hipFree(d_cig), (void)hipFree(d_out), (void)hipFree(d_all_out), (void)hipFree(d_all_cig);
    sdf_destroy(ctx);
  };
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r);
  for (auto &t : th) t.join();
  for (auto c : comms) sdf_comm_destroy(c);
  int64_t cells = 0, cmax = 0;
  for (int r = 0; r < world; ++r) cells += shard_cells[r], cmax = std::max(cmax, shard_cells[r]);
  const double tmax = *std::max_element(secs.begin(), secs.end());
  printf("{\"metric\": \"aligned DP cells/sec (Gcell/s), one batch sharded over the devices + RCCL all-gatherv of the results\", "
         "\"value\": %.3f, \"unit\": \"Gcell/s\", \"n_gpus\": %d, \"steps\": %d, \"ms_per_step\": %.3f, \"tasks\": %zu, "
         "\"shard_balance_max_over_mean\": %.4f, \"union_check\": \"%s\", \"host\": \"C++ (sdf_multi), one thread per device, "
         "ncclCommInitAll\"}\n",
         (double)cells * steps / tmax / 1e9, world, steps, tmax / steps * 1e3, n, (double)cmax * world / (double)std::max<int64_t>(cells, 1),
         failures.load() ? "FAILED" : "every device's part equals what it computed; the union holds one complete record per task");
  for (auto &s : notes)
    if (!s.empty()) fprintf(stderr, "sdf_multi: %s\n", s.c_str());
  return failures.load() ? 1 : 0;
}
