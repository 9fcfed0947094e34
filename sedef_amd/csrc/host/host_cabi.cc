// C ABI of the host pipeline (used by the CLI wrapper tests and by the Python tests through ctypes).
// `test_dp` arguments are TEST HOOKS: a non-NULL function with the oracle's single-task signature
// replaces the GPU provider; the CLI always passes NULL.
#include <cstdio>
#include <cstring>
#include <string>

#include "sedef_host.h"

using namespace sdfh;

namespace sdfh { void set_alignment_scoring(const Params &p); }

static thread_local std::string g_err;

static std::unique_ptr<DpProvider> provider(test_dp_fn fn, int device) {
  set_stage_settings(StageSettings::from_env());  // (every run of an entry point: the stage driver's settings, read here once)
  if (fn) return make_test_provider(fn);
  return make_gpu_provider(device);
}

static int copy_out(const std::string &s, char *buf, size_t cap) {
  if (s.size() + 1 > cap) {
    g_err = "output buffer too small";
    return -2;
  }
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

extern "C" {

const char *sdfh_last_error(void) { return g_err.c_str(); }

// `sedef align generate -k K genome.fa bucket.bed > out` (reference: src/align_main.cc:285-337)
int sdfh_generate(const char *ref_path, const char *bed_path, int kmer, const char *out_path, int match,
                  int mismatch, int gap_open, int gap_extend, test_dp_fn test_dp, int device, long long *stats) {
  try {
    Params p;
    p.match = match;
    p.mismatch = mismatch;
    p.gap_open = gap_open;
    p.gap_extend = gap_extend;
    auto dp = provider(test_dp, device);
    FILE *out = out_path ? fopen(out_path, "w") : stdout;
    if (!out) throw std::string("Cannot open output ") + out_path;
    GenerateStats st = generate_alignments(ref_path, bed_path, kmer, p, *dp, out, stderr);
    if (out_path) fclose(out);
    if (stats) {
      stats[0] = st.lines;
      stats[1] = st.total_written;
      stats[2] = st.dp_tasks;
      stats[3] = st.dp_cells;
      stats[4] = st.rounds;
    }
    return 0;
  } catch (std::string &s) {
    g_err = s;
    return -1;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// several buckets, one provider (host/pipeline.cc: generate_many): `beds` = bucket paths separated by newlines; every
// bucket's lines go to `<bucket><out_suffix>`.  stats: per bucket {lines, hits} (2 x buckets entries).
int sdfh_generate_many(const char *ref_path, const char *beds, int kmer, const char *out_suffix, const char *log_dir,
                       test_dp_fn test_dp, int device, long long *stats) {
  try {
    Params p;
    auto dp = provider(test_dp, device);
    std::vector<std::string> list;
    for (const char *c = beds; *c;) {
      const char *e = strchr(c, '\n');
      std::string one = e ? std::string(c, e) : std::string(c);
      if (!one.empty()) list.push_back(one);
      if (!e) break;
      c = e + 1;
    }
    const auto sts = generate_many(ref_path, expand_buckets(list), kmer, p, *dp, out_suffix, log_dir ? log_dir : "", stderr);
    for (size_t k = 0; stats && k < sts.size(); k++) stats[2 * k] = sts[k].lines, stats[2 * k + 1] = sts[k].total_written;
    return (int)sts.size();
  } catch (std::string &s) {
    g_err = s;
    return -1;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// `sedef stats generate genome.fa final.bed > out` (reference: src/stats_main.cc:339-389); test_cols: the oracle's column
// walker instead of the device (CPU tests).  stats: hits read, pieces, columns; returns the lines written or -1.
long sdfh_stats_generate(const char *ref_path, const char *bed_path, const char *out_path, int max_ok_gap, int min_split,
                         int min_uppercase, double max_error, test_cols_fn test_cols, int device, long long *stats) {
  try {
    StatsParams sp;
    sp.max_ok_gap = max_ok_gap;
    sp.min_split = min_split;
    sp.min_uppercase = min_uppercase;
    sp.max_scaled_error = max_error;
    struct Out {  // (closed on every path; a run that throws leaves no truncated table behind -- ADVICE r3)
      FILE *f = nullptr;
      std::string path;
      bool done = false;
      ~Out() {
        if (f) fclose(f);
        if (f && !done) remove(path.c_str());
      }
    } out;
    out.path = out_path;
    out.f = fopen(out_path, "w");
    if (!out.f) throw std::string("Cannot open output ") + out_path;
    const long lines = stats_generate(ref_path, bed_path, out.f, sp, test_cols, device, stats);
    out.done = true;
    return lines;
  } catch (std::string &s) {
    g_err = s;
    return -1;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// fmt 4.0.1 "{}" of a double, as `stats generate` prints it (test hook)
int sdfh_format_double(double x, char *buf, size_t cap) { return copy_out(format_double(x), buf, cap); }

// Alignment(fa, fb) (reference: src/align.cc:76-88): CIGAR string + counters {matches, mismatches, gaps, gap_bases, span}
static int alignment_pair_impl(const Params &p, const char *fa, const char *fb, test_dp_fn test_dp, int device,
                               char *cigar, size_t cap, int *counts);
int sdfh_alignment_pair(const char *fa, const char *fb, test_dp_fn test_dp, int device, char *cigar, size_t cap,
                        int *counts) {
  return alignment_pair_impl(Params(), fa, fb, test_dp, device, cigar, cap, counts);
}
// the same with the CLI's scoring overrides (reference: src/align_main.cc:343-352)
int sdfh_alignment_pair_scored(const char *fa, const char *fb, int match, int mismatch, int gap_open, int gap_extend,
                               test_dp_fn test_dp, int device, char *cigar, size_t cap, int *counts) {
  Params p;
  p.match = match;
  p.mismatch = mismatch;
  p.gap_open = gap_open;
  p.gap_extend = gap_extend;
  const int rc = alignment_pair_impl(p, fa, fb, test_dp, device, cigar, cap, counts);
  set_alignment_scoring(Params());
  return rc;
}
static int alignment_pair_impl(const Params &p, const char *fa, const char *fb, test_dp_fn test_dp, int device,
                               char *cigar, size_t cap, int *counts) {
  try {
    set_alignment_scoring(p);
    auto dp = provider(test_dp, device);
    const std::string sa = fa, sb = fb;  // (requests point into the sequences: they live until the DP has run)
    std::vector<DpRequest> reqs;
    DpSession rec;
    rec.requests = &reqs;
    { Alignment tmp(sa, sb, rec); }
    std::vector<Cigar> res = dp->run(reqs, p);
    DpSession rep;
    rep.recording = false;
    rep.results = &res;
    Alignment al(sa, sb, rep);
    counts[0] = al.matches();
    counts[1] = al.mismatches();
    counts[2] = al.gaps();
    counts[3] = al.gap_bases();
    counts[4] = al.span();
    return copy_out(al.cigar_string(), cigar, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// Guide alignment of refined chains (reference: src/align.cc:107-197 after optional merges, src/refine.cc:165-183).
// Hits are given as coordinates + CIGAR strings over qstr/rstr; consecutive overlapping hits are merged
// like refine_chains does.  Output: "qs qe rs re cigar matches mismatches gaps gap_bases".
int sdfh_guide_alignment(const char *qstr_, const char *rstr_, int n, const int *coords, const char **cigars, int side,
                         test_dp_fn test_dp, int device, char *outbuf, size_t cap) {
  try {
    Params p;
    set_alignment_scoring(p);
    auto dp = provider(test_dp, device);
    const std::string qstr = qstr_, rstr = rstr_;
    auto qs = std::make_shared<Sequence>("QRY", qstr);
    auto rs = std::make_shared<Sequence>("REF", rstr);
    std::vector<Hit> hits(n);
    for (int k = 0; k < n; k++) {
      Hit &h = hits[k];
      h.query = qs;
      h.ref = rs;
      h.query_start = coords[4 * k];
      h.query_end = coords[4 * k + 1];
      h.ref_start = coords[4 * k + 2];
      h.ref_end = coords[4 * k + 3];
      h.aln = Alignment(qstr.substr(h.query_start, h.query_end - h.query_start),
                        rstr.substr(h.ref_start, h.ref_end - h.ref_start), std::string(cigars[k]));
      h.aln.start_a = h.query_start;
      h.aln.end_a = h.query_end;
      h.aln.start_b = h.ref_start;
      h.aln.end_b = h.ref_end;
    }
    auto with_dp = [&](std::function<void(DpSession &)> body_rec, std::function<void(DpSession &)> body_real) {
      std::vector<DpRequest> reqs;
      DpSession rec;
      rec.requests = &reqs;
      body_rec(rec);
      std::vector<Cigar> res = dp->run(reqs, p);
      DpSession rep;
      rep.recording = false;
      rep.results = &res;
      body_real(rep);
    };
    std::vector<Hit> guide;
    Hit *prev = &hits[0];
    for (int pi = 1; pi < n; pi++) {
      Hit &cur = hits[pi];
      if (cur.query_start < prev->query_end || cur.ref_start < prev->ref_end) {
        with_dp([&](DpSession &d) { Alignment a = prev->aln, c = cur.aln; a.merge(c, qstr, rstr, d); },
                [&](DpSession &d) { prev->aln.merge(cur.aln, qstr, rstr, d); });
        update_from_alignment(*prev);
      } else {
        guide.push_back(*prev);
        prev = &cur;
      }
    }
    guide.push_back(*prev);
    Alignment fin;
    with_dp([&](DpSession &d) { Alignment t(qstr, rstr, guide, side, d); },
            [&](DpSession &d) { fin = Alignment(qstr, rstr, guide, side, d); });
    char head[256];
    snprintf(head, sizeof head, "%d %d %d %d ", fin.start_a, fin.end_a, fin.start_b, fin.end_b);
    char tail[128];
    snprintf(tail, sizeof tail, " %d %d %d %d", fin.matches(), fin.mismatches(), fin.gaps(), fin.gap_bases());
    return copy_out(std::string(head) + fin.cigar_string() + tail, outbuf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// Chains given as anchor lists "q,r,l;q,r,l|q,r,l;...": builds every chain alignment (src/align.cc:199-270),
// merges overlapping neighbours and builds the guide alignment with side extension, like refine_chains does for
// one path (src/refine.cc:163-183).  Output: "qs qe rs re cigar matches mismatches gaps gap_bases|to_bed line".
int sdfh_guide_from_chains(const char *qstr_, const char *rstr_, const char *spec, int side, test_dp_fn test_dp,
                           int device, char *outbuf, size_t cap) {
  try {
    Params p;
    set_alignment_scoring(p);
    auto dp = provider(test_dp, device);
    const std::string qstr = qstr_, rstr = rstr_;
    std::vector<std::vector<Anchor>> chains(1);
    for (const char *c = spec; *c;) {
      if (*c == '|') {
        chains.push_back({});
        c++;
        continue;
      }
      int q, r, l, n = 0;
      if (sscanf(c, "%d,%d,%d%n", &q, &r, &l, &n) != 3) break;
      chains.back().push_back(Anchor{q, r, l, 0});
      c += n;
      if (*c == ';') c++;
    }
    if (chains.back().empty()) chains.pop_back();
    auto with_dp = [&](std::function<void(DpSession &)> body) {
      std::vector<DpRequest> reqs;
      DpSession rec;
      rec.requests = &reqs;
      body(rec);  // recording pass (callers pass bodies that work on copies)
      return dp->run(reqs, p);
    };
    auto qs = std::make_shared<Sequence>("QRY", qstr);
    auto rs = std::make_shared<Sequence>("REF", rstr);
    std::vector<Hit> hits(chains.size());
    for (size_t k = 0; k < chains.size(); k++) {
      std::vector<int> idx;
      for (int i = 0; i < (int)chains[k].size(); i++) idx.push_back(i);
      auto res = with_dp([&](DpSession &d) { Alignment t(qstr, rstr, chains[k], idx, d); });
      DpSession rep;
      rep.recording = false;
      rep.results = &res;
      hits[k].query = qs;
      hits[k].ref = rs;
      hits[k].aln = Alignment(qstr, rstr, chains[k], idx, rep);
      update_from_alignment(hits[k]);
    }
    std::vector<Hit> guide;
    Hit *prev = &hits[0];
    for (size_t pi = 1; pi < hits.size(); pi++) {
      Hit &cur = hits[pi];
      if (cur.query_start < prev->query_end || cur.ref_start < prev->ref_end) {
        auto res = with_dp([&](DpSession &d) { Alignment a = prev->aln, c = cur.aln; a.merge(c, qstr, rstr, d); });
        DpSession rep;
        rep.recording = false;
        rep.results = &res;
        prev->aln.merge(cur.aln, qstr, rstr, rep);
        update_from_alignment(*prev);
      } else {
        guide.push_back(*prev);
        prev = &cur;
      }
    }
    guide.push_back(*prev);
    auto res = with_dp([&](DpSession &d) { Alignment t(qstr, rstr, guide, side, d); });
    DpSession rep;
    rep.recording = false;
    rep.results = &res;
    Hit fin;
    fin.query = qs;
    fin.ref = rs;
    fin.name = "name";
    fin.comment = "cmt";
    fin.aln = Alignment(qstr, rstr, guide, side, rep);
    update_from_alignment(fin);
    char head[256], tail[128];
    snprintf(head, sizeof head, "%d %d %d %d ", fin.query_start, fin.query_end, fin.ref_start, fin.ref_end);
    snprintf(tail, sizeof tail, " %d %d %d %d|", fin.aln.matches(), fin.aln.mismatches(), fin.aln.gaps(),
             fin.aln.gap_bases());
    return copy_out(std::string(head) + fin.aln.cigar_string() + tail + fin.to_bed(false), outbuf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// fast_align(query, ref, orig, kmer) (reference: src/chain.cc:203-268, the per-pair entry src/align_main.cc:313 and
// python/sedef.cpp:85 call): seed anchors, chains, chain alignments, refine_chains.  `orig` is given by the fields the
// path reads: the two names and strands (same-chromosome rules, src/chain.cc:45-46, src/refine.cc:29-30) and the two
// start coordinates.  Output: one line per refined hit, coordinates relative to the two strings:
// "qs qe rs re cigar matches mismatches gaps gap_bases\n".
int sdfh_fast_align(const char *query_, const char *ref_, const char *qname, const char *rname, int q_rc, int r_rc,
                    int qstart, int rstart, int kmer, test_dp_fn test_dp, int device, char *outbuf, size_t cap) {
  try {
    Params p;
    p.kmer = kmer;
    set_alignment_scoring(p);
    auto dp = provider(test_dp, device);
    const std::string query = query_, ref = ref_;
    Hit orig;
    orig.query = std::make_shared<Sequence>(qname, "", false);
    orig.ref = std::make_shared<Sequence>(rname, "", false);
    orig.query->is_rc = q_rc != 0;
    orig.ref->is_rc = r_rc != 0;
    orig.query_start = qstart;
    orig.ref_start = rstart;
    orig.query_end = qstart + (int)query.size();
    orig.ref_end = rstart + (int)ref.size();
    PairJob job(query, ref, orig, p);
    std::vector<Cigar> results;
    for (;;) {
      std::vector<DpRequest> reqs = job.advance(results);
      if (reqs.empty()) break;
      results = dp->run(reqs, p);
    }
    std::string out;
    char nums[160];
    for (auto &h : job.hits()) {
      snprintf(nums, sizeof nums, "%d %d %d %d ", h.query_start, h.query_end, h.ref_start, h.ref_end);
      out += nums;
      out += h.aln.cigar_string();
      snprintf(nums, sizeof nums, " %d %d %d %d\n", h.aln.matches(), h.aln.mismatches(), h.aln.gaps(), h.aln.gap_bases());
      out += nums;
    }
    return copy_out(out, outbuf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  } catch (std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// FastaReference::get_sequence (reference: src/fasta.cc:105-142); *end is updated like the reference does
int sdfh_fasta_get(const char *path, const char *name, int start, int *end, char *buf, size_t cap) {
  try {
    FastaReference fr(path);
    std::string s = fr.get_sequence(name, start, end);
    return copy_out(s, buf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// merge() of src/merge.cc:35-109 on BED lines (one per line in `lines`); output: merged BED lines
int sdfh_merge(const char *lines, int merge_dist, char *buf, size_t cap) {
  try {
    std::vector<Hit> hits;
    for (auto &l : split(lines, '\n'))
      if (!l.empty()) hits.push_back(Hit::from_bed(l));
    auto res = merge_hits(hits, merge_dist);
    std::string out;
    for (auto &h : res) out += h.to_bed(false) + "\n";
    return copy_out(out, buf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// `sedef align bucket -n N bed_path out_dir genome.fa` (reference: src/align_main.cc:38-198)
int sdfh_bucket(const char *bed_path, int nbins, const char *out_dir, const char *reference) {
  try {
    BucketParams bp;
    bucket_alignments_extern(bed_path, nbins, out_dir, true, reference, bp, stderr);
    return 0;
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// generate_anchors on the host (reference: src/chain.cc:24-101): "q r l has_u;" per anchor, in order
int sdfh_anchors(const char *query, const char *ref, int kmer, int same_chr, int qstart, int rstart, char *buf,
                 size_t cap) {
  try {
    Hit orig;
    orig.query = std::make_shared<Sequence>("A", "");
    orig.ref = std::make_shared<Sequence>(same_chr ? "A" : "B", "");
    orig.query_start = qstart;
    orig.ref_start = rstart;
    std::vector<Anchor> an = generate_anchors(query, ref, orig, kmer);
    std::string out;
    char t[96];
    for (auto &a : an) {
      snprintf(t, sizeof t, "%d %d %d %d;", a.q, a.r, a.l, a.has_u);
      out += t;
    }
    return copy_out(out, buf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// Chain extraction for one pair: anchors + chains (reference: src/chain.cc:24-199).  Writes
// "q r l has_u" per anchor of each kept chain, chains separated by "|" -- for self-consistency tests.
int sdfh_chains(const char *query, const char *ref, int kmer, char *buf, size_t cap) {
  try {
    Params p;
    p.kmer = kmer;
    Hit orig;
    orig.query = std::make_shared<Sequence>("A", "");
    orig.ref = std::make_shared<Sequence>("B", "");
    std::vector<Anchor> an = generate_anchors(query, ref, orig, kmer);
    auto ch = chain_anchors(an, p);
    std::string out;
    for (size_t bi = 1; bi < ch.second.size(); bi++) {
      for (int k = ch.second[bi].first - 1; k >= ch.second[bi - 1].first; k--) {
        const Anchor &a = an[ch.first[k]];
        char t[96];
        snprintf(t, sizeof t, "%d %d %d %d;", a.q, a.r, a.l, a.has_u);
        out += t;
      }
      out += "|";
    }
    return copy_out(out, buf, cap);
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// chain_anchors on a caller-provided anchor list: path (m entries), bounds (2*(m+1)), returns #boundaries
int sdfh_chain_raw(const int32_t *anchors, int m, int max_chain_gap, int match_chain_score, int32_t *path,
                   int32_t *bounds) {
  try {
    Params p;
    p.max_chain_gap = max_chain_gap;
    p.match_chain_score = match_chain_score;
    std::vector<Anchor> an(m);
    for (int i = 0; i < m; i++) an[i] = Anchor{anchors[4 * i], anchors[4 * i + 1], anchors[4 * i + 2], anchors[4 * i + 3]};
    auto ch = chain_anchors(an, p);
    for (size_t k = 0; k < ch.first.size(); k++) path[k] = ch.first[k];
    for (size_t b = 0; b < ch.second.size(); b++) {
      bounds[2 * b] = ch.second[b].first;
      bounds[2 * b + 1] = ch.second[b].second ? 1 : 0;
    }
    return (int)ch.second.size();
  } catch (std::string &s) {
    g_err = s;
    return -1;
  }
}

// Test hook: the priority search tree of chain_anchors driven by a script (see chain.cc: rangemax_script)
int sdfh_rangemax_script(const int *pts, int n, const int *ops, int nops, int *out, int *state, int state_cap) {
  return rangemax_script(pts, n, ops, nops, out, state, state_cap);
}

// Hit::extend (reference: src/hit.cc:200-207): io = {query_start, query_end, ref_start, ref_end}
int sdfh_hit_extend(int *io, double factor, int max_extend) {
  Hit h;
  h.query_start = io[0];
  h.query_end = io[1];
  h.ref_start = io[2];
  h.ref_end = io[3];
  h.extend(factor, max_extend);
  io[0] = h.query_start;
  io[1] = h.query_end;
  io[2] = h.ref_start;
  io[3] = h.ref_end;
  return 0;
}

// Sequence ctor (reference: src/hash.cc:104-109): "name|seq|is_rc"
int sdfh_sequence(const char *name, const char *seq, int is_rc, char *buf, size_t cap) {
  Sequence s(name, seq, is_rc != 0);
  return copy_out(s.name + "|" + s.seq + "|" + (s.is_rc ? "1" : "0"), buf, cap);
}

}  // extern "C"
