// Host-side profile of the stage driver WITHOUT a device: the product's host code (chaining, stitching, refinement,
// output) on a DP hook that answers at once (min(q,t) M + the rest as one gap run -- not an alignment, only a workload
// of the right shape), so that gprof / the stage's own clocks show where the HOST time goes.
//   g++ -O2 -pg -g ... (profiles/r06_host_prof.sh)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>

#include "../sedef_amd/csrc/host/sedef_host.h"

struct FakeRes {  // layout of sdfo_result (oracle/extz2_oracle.h)
  uint32_t max;
  int32_t zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score;
  int64_t n_cigar;
  uint32_t *cigar;
};
static void fake_dp(int qlen, const uint8_t *, int tlen, const uint8_t *, int, const int8_t *, int, int, int, int, int, void *out) {
  FakeRes *r = (FakeRes *)out;
  const int m = qlen < tlen ? qlen : tlen;
  r->cigar = (uint32_t *)malloc(8);
  r->n_cigar = 0;
  if (m) r->cigar[r->n_cigar++] = (uint32_t)m << 4;
  if (qlen > m) r->cigar[r->n_cigar++] = (uint32_t)(qlen - m) << 4 | 1;
  if (tlen > m) r->cigar[r->n_cigar++] = (uint32_t)(tlen - m) << 4 | 2;
}
// the hook's provider plus seed anchors "from the device": computed once with the host's generate_anchors (all host threads,
// outside the jobs' clocks) and handed to the driver like sdf_anchors_batch's output, so that the jobs' thread time is what
// the GPU path's jobs spend (chaining, recording, stitching) and not 16 s of host k-mer lookup
struct ProfProvider : sdfh::DpProvider {
  std::unique_ptr<sdfh::DpProvider> inner = sdfh::make_test_provider(fake_dp);
  std::vector<std::vector<sdfh::Anchor>> cache;
  std::vector<sdfh::Cigar> run(const std::vector<sdfh::DpRequest> &reqs, const sdfh::Params &p) override {
    auto r = inner->run(reqs, p);
    tasks = inner->tasks, cells = inner->cells;
    return r;
  }
  bool anchors(const std::vector<AnchorJob> &jobs, int kmer, AnchorBatch &out) override {
    if (cache.size() != jobs.size()) {
      cache.assign(jobs.size(), {});
      std::vector<std::thread> thr;
      std::atomic<size_t> next(0);
      for (int t = 0; t < 8; t++)
        thr.emplace_back([&] {
          for (size_t k = next++; k < jobs.size(); k = next++) {
            sdfh::Hit h;
            h.query = std::make_shared<sdfh::Sequence>(jobs[k].same_chr ? "a" : "a", "");
            h.ref = std::make_shared<sdfh::Sequence>(jobs[k].same_chr ? "a" : "b", "");
            h.query_start = 0;
            h.ref_start = jobs[k].delta;
            cache[k] = sdfh::generate_anchors(jobs[k].query.str(), jobs[k].ref.str(), h, kmer);
          }
        });
      for (auto &t : thr) t.join();
    }
    out.off.assign(jobs.size() + 1, 0);
    for (size_t k = 0; k < jobs.size(); k++) out.off[k + 1] = out.off[k] + (int64_t)cache[k].size();
    out.buf.reset(new sdfh::Anchor[(size_t)out.off.back() + 1]);
    for (size_t k = 0; k < jobs.size(); k++) std::copy(cache[k].begin(), cache[k].end(), out.buf.get() + out.off[k]);
    return true;
  }
};

int main(int argc, char **argv) {
  if (argc < 3) return 1;
  sdfh::set_stage_settings(sdfh::StageSettings::from_env());
  std::unique_ptr<sdfh::DpProvider> dp(new ProfProvider);
  sdfh::Params p;
  FILE *out = fopen("/dev/null", "w");
  try {
    for (int it = 0; it < (argc > 3 ? atoi(argv[3]) : 1); it++) sdfh::generate_alignments(argv[1], argv[2], 11, p, *dp, out, stderr);
  } catch (std::string &s) {
    fprintf(stderr, "Error: %s\n", s.c_str());
    return 1;
  }
  return 0;
}
