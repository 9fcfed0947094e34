// `sedef` command line for the align stage (restates the CLI contract of reference src/main.cc:104-157 and
// src/align_main.cc:341-373): `sedef align generate -k K [--match N --mismatch N --gap-open N --gap-extend N]
// genome.fa bucket.bed`.  stdout carries only the BEDPE lines; everything else goes to stderr.
// The DP runs on the GPU; without a HIP device the command fails with exit code 1.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <malloc.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sedef_host.h"

using namespace sdfh;

namespace {
// argh PREFER_PARAM_FOR_UNREG_OPTION (reference: extern/argh.h:211-232): any -x/--x followed by a token that
// is not itself an option takes it as its value (negative numbers are values); the rest are positionals.
struct Args {
  std::vector<std::string> pos;
  std::vector<std::pair<std::string, std::string>> params;
  static bool is_number(const std::string &s) {
    char *e = nullptr;
    strtod(s.c_str(), &e);
    return e && *e == 0 && !s.empty();
  }
  static bool is_option(const std::string &s) { return s.size() > 1 && s[0] == '-' && !is_number(s); }
  Args(int argc, char **argv) {
    for (int i = 0; i < argc; i++) {
      std::string a = argv[i];
      if (!is_option(a)) {
        pos.push_back(a);
        continue;
      }
      std::string name = a.substr(a.find_first_not_of('-'));
      const size_t eq = name.find('=');
      if (eq != std::string::npos) {
        params.push_back({name.substr(0, eq), name.substr(eq + 1)});
      } else if (i + 1 < argc && !is_option(argv[i + 1])) {
        params.push_back({name, argv[++i]});
      }
    }
  }
  bool getd(std::initializer_list<const char *> names, double &v) const {
    for (auto &p : params)
      for (auto n : names)
        if (p.first == n) {
          v = atof(p.second.c_str());
          return true;
        }
    return false;
  }
  bool get(std::initializer_list<const char *> names, int &v) const {
    for (auto &p : params)
      for (auto n : names)
        if (p.first == n) {
          v = atoi(p.second.c_str());
          return true;
        }
    return false;
  }
};
}  // namespace

// A crash says where: the frames of the faulting thread on stderr (the reference dies silently; sedef.sh:195 only counts the
// "Finished" lines that are missing afterwards).
static void crash_handler(int sig) {
  void *frames[48];
  const int n = backtrace(frames, 48);
  const char msg[] = "\nsedef: fatal signal, frames of the faulting thread:\n";
  if (write(2, msg, sizeof(msg) - 1) < 0) _exit(128 + sig);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

int main(int argc, char **argv) {
  const auto t_main = std::chrono::steady_clock::now();
  signal(SIGSEGV, crash_handler);
  signal(SIGBUS, crash_handler);
  signal(SIGABRT, crash_handler);
  // the DP path keeps four streams busy; give the HIP runtime more hardware queues than its default of four so
  // that no two of them share one (has to be in the environment before the runtime initialises)
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  // Copies by shader kernels, not by the SDMA engines: the FIRST device-to-host copy of a stage run that goes through SDMA costs
  // its caller 8-16 ms on this hardware (the anchors' way back or the first round's results, whichever comes first; the engine
  // has to be woken), a blit kernel 0.4 ms -- chr1-sized bucket 0.10 -> 0.08 s (profiles/r06_sdma_probe.txt).  Before the
  // runtime starts; a value the user has set stays.
  setenv("HSA_ENABLE_SDMA", "0", 0);
  // the per-pair host work allocates and frees megabytes on every thread: keep freed memory in the arenas instead
  // of returning it to the kernel each time (munmap / page faults serialise the threads on the address-space lock)
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
  mallopt(M_TOP_PAD, 256 << 20);
  if (argc < 2) {
    fprintf(stderr, "Arguments missing: please run sedef help for more information.\n");
    return 1;
  }
  fprintf(stderr, "SEDEF align stage on MI355X (sedef_amd); arguments: ");
  for (int i = 0; i < argc; i++) fprintf(stderr, " %s", argv[i]);
  fprintf(stderr, "\n");
  const std::string command = argv[1];
  if (argc < 3 && command != "help") {
    fprintf(stderr, "Arguments missing: please run sedef help for more information.\n");
    return 1;
  }
  try {
    set_stage_settings(StageSettings::from_env());  // (SDF_LANES, SDF_DEVICES, SDF_SUPER_BATCH, ...: read here, once; sedef_host.h)
    if (command == "help") {
      fprintf(stderr,
              "sedef align generate -k [kmer] [genome.fa] [initial.bed]\n"
              "  generates true alignments for [initial.bed] (BEDPE on stdout)\n"
              "sedef align generate -k [kmer] [--log-dir dir] [genome.fa] [bucket ... | bucket directory]\n"
              "  several buckets in one process: bucket b's BEDPE goes to b.aligned.bed, its log to dir/b.log\n"
              "  params: -k/--kmer, --match, --mismatch, --gap-open, --gap-extend (default 5, -4, -40, -1)\n"
              "sedef align bucket -n [count] [bed_directory(/)] [buckets/] [genome.fa]\n"
              "  bucket BEDs into [count] files for the alignment stage (--extend-ratio, --max-extend, --merge-dist)\n"
              "sedef stats generate [genome.fa] [final.bed]\n"
              "  the per-alignment table of the final calls (--max-ok-gap, --min-split, --uppercase, --max-error)\n"
              "Other SEDEF stages (search, stats diff, translate) are not part of this build.\n");
      return 0;
    } else if (command == "align") {
      Args a(argc - 2, argv + 2);
      Params p;
      a.get({"match"}, p.match);
      a.get({"mismatch"}, p.mismatch);
      a.get({"gap-open"}, p.gap_open);
      a.get({"gap-extend"}, p.gap_extend);
      if (a.pos.size() < 3) throw std::string("Not enough arguments to align");
      if (a.pos[0] == "generate") {
        int k;
        if (!a.get({"k", "kmer"}, k)) throw std::string("Must provide k-mer size (--kmer)");
        const int dv = stage_settings().device;
        const auto t0 = std::chrono::steady_clock::now();
        // (the lanes of the stage driver and the size of its super-batches follow from the seed pairs: the lanes' device
        // contexts and buffers are set up side by side here, not one after the other inside the stage)
        // `generate genome.fa bucket`: the reference's form, BEDPE on stdout.  With several bucket files, or a directory of
        // `bucket_????` files, ONE process aligns them all: bucket b's lines go to `b.aligned.bed` (where sedef.sh:189 redirects
        // that bucket's stdout), its "Finished BED" line to stderr and, with --log-dir, to `<dir>/<bucket>.log` (sedef.sh:195
        // counts them) -- device contexts, lanes and buffers set up once instead of once per bucket.
        std::vector<std::string> given(a.pos.begin() + 2, a.pos.end());
        const std::vector<std::string> buckets = expand_buckets(given);
        const bool many = buckets.size() != 1 || buckets[0] != given[0];
        std::string log_dir, suffix = ".aligned.bed";
        for (auto &pr : a.params) {
          if (pr.first == "log-dir") log_dir = pr.second;
          if (pr.first == "out-suffix") suffix = pr.second;
        }
        if (buckets.empty()) throw std::string("No bucket files in ") + given[0];
        const StageHint hint = many ? stage_hint_many(buckets) : stage_hint(a.pos[2]);
        if (!many && stage_settings().stage_ws_gib <= 0) {
          // One bucket per process (sedef.sh:187-190) means one such process right after the other, and a process that follows
          // one which held 16 GiB of direction flags pays for it: context 0.45 s against 0.33, two runs in eight 0.5 s more on
          // top (the device buffers' allocation waits), where the larger workspace saves its own stage 8 ms
          // (profiles/r06_proc_probe.txt: wall 0.73-0.75 s with outliers of 1.2 against 0.61-0.67).  Sixteen stay for a
          // process that aligns many buckets.
          StageSettings s = stage_settings();
          s.stage_ws_gib = 8;
          set_stage_settings(s);
        }
        auto dp = make_gpu_providers(dv, hint.lanes, hint.devices, hint.max_batch_bytes);
        const auto t1 = std::chrono::steady_clock::now();
        if (many) {
          const auto sts = generate_many(a.pos[1], buckets, k, p, *dp, suffix, log_dir, stderr);
          long long lines = 0, hits = 0;
          for (auto &x : sts) lines += x.lines, hits += x.total_written;
          fprintf(stderr, "All %zu buckets done in %.2fs (%lld lines, generated %lld hits)\n", sts.size(),
                  std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(), lines, hits);
        } else {
          generate_alignments(a.pos[1], a.pos[2], k, p, *dp, stdout, stderr);
        }
        const auto t2 = std::chrono::steady_clock::now();
        dp.reset();
        if (stage_settings().debug_timing)
          fprintf(stderr, "  [process: main to context %.3fs, device context %.3fs, stage %.3fs, context teardown %.3fs]\n",
                  std::chrono::duration<double>(t0 - t_main).count(),
                  std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(),
                  std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count());
      } else if (a.pos[0] == "bucket") {
        int nbins;
        if (!a.get({"n", "bins"}, nbins)) throw std::string("Must provide number of bins (--bins)");
        BucketParams bp;
        a.getd({"extend-ratio"}, bp.extend_ratio);
        a.get({"max-extend"}, bp.max_extend);
        a.get({"merge-dist"}, bp.merge_dist);
        if (a.pos.size() < 4) throw std::string("Not enough arguments to align");
        bucket_alignments_extern(a.pos[1], nbins, a.pos[2], true, a.pos[3], bp, stderr);
      } else {
        throw std::string("Unknown align command");
      }
    } else if (command == "stats") {
      // `sedef stats generate [--max-ok-gap N --min-split N --uppercase N --max-error X] genome.fa final.bed`
      // (reference: src/stats_main.cc:482-510); `stats diff` (WGAC comparison) is not part of this build
      Args a(argc - 2, argv + 2);
      StatsParams sp;
      a.get({"max-ok-gap"}, sp.max_ok_gap);
      a.get({"min-split"}, sp.min_split);
      a.get({"uppercase"}, sp.min_uppercase);
      a.getd({"max-error"}, sp.max_scaled_error);
      if (a.pos.size() < 3) throw std::string("Not enough arguments to stats");
      if (a.pos[0] != "generate") throw std::string("Unknown stats command");
      long long st[3] = {0, 0, 0};
      const long lines = stats_generate(a.pos[1], a.pos[2], stdout, sp, nullptr, stage_settings().device, st);
      fprintf(stderr, "Processed hit %lld out of %lld... done! (%lld pieces, %lld columns on the device, %ld lines)\n", st[0], st[0],
              st[1], st[2], lines);
    } else {
      fprintf(stderr, "Whoops, invalid command!\n");
    }
  } catch (std::string &s) {
    fprintf(stderr, "Error: %s\n", s.c_str());
    fprintf(stderr, "Double-check the parameters: run sedef --help for explanation.\n");
    return 1;
  } catch (std::exception &e) {
    fprintf(stderr, "Error: %s\n", e.what());
    return 1;
  }
  return 0;
}
