// Host-side profile of the stage driver WITHOUT a device: the product's host code (chaining, stitching, refinement,
// output) on a DP hook that answers at once (min(q,t) M + the rest as one gap run -- not an alignment, only a workload
// of the right shape), so that gprof / the stage's own clocks show where the HOST time goes.
//   g++ -O2 -pg -g ... (profiles/r06_host_prof.sh)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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
int main(int argc, char **argv) {
  if (argc < 3) return 1;
  sdfh::set_stage_settings(sdfh::StageSettings::from_env());
  auto dp = sdfh::make_test_provider(fake_dp);
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
