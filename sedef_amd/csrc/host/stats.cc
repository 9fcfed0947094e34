// `sedef stats generate` (reference: src/stats_main.cc:33-336), the immediate consumer of `align generate`'s BEDPE --
// scope row f4, host side.  The reference expands every alignment into three column strings and walks them three times
// (assembly-gap search, trims, the counter loop).  Here an alignment stays a run-length CIGAR over the two FASTA strings
// (alignment.cc); the per-column counters of ALL pieces of a file are taken in one call of sdf_stats_columns_batch
// (stats_cols.hip) -- the device reads the characters and the runs, nothing is expanded -- and this file does the text
// around it: BEDPE parsing, the query / reference swap, the order, the cuts at assembly gaps and large gaps
// (split_alignment / gap_split / subhit), the four floating-point columns, the filters and the formatting.
//
// Parity: src/stats_main.cc includes boost/dynamic_bitset.hpp and cannot be compiled here -- this file is a restatement,
// parity unpinned, except for what it shares with pinned pieces: Alignment(fa, fb, cigar), trim_front / trim_back,
// Hit::from_bed / to_bed (tests/test_pinning.py, tests/test_host_pipeline.py), the number formatting (fmt 4.0.1 "{}" of a
// double is printf's %g: tests/test_stats_generate.py against the reference's vendored fmt) and the column counters
// (tests/test_stats_columns.py).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <tuple>

#include "../../../include/sedef_hip.h"
#include "sedef_host.h"

namespace sdfh {

std::string format_double(double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%g", x);
  return buf;
}

namespace {
const int kMinAssemblyGap = 100;   // Globals::Stats::MIN_ASSEMBLY_GAP_SIZE (src/globals.h:101)
const int kBigOverlap = 100;       // Globals::Stats::BIG_OVERLAP_THRESHOLD (src/globals.h:102)

// What subhit keeps of a Hit: coordinates, names, strands and the alignment (src/stats_main.cc:33-84).
struct Piece {
  std::shared_ptr<Sequence> query, ref;
  int query_start, query_end, ref_start, ref_end;
  Alignment aln;
};

// columns [start, end) of hin as a hit of their own.  The hit's coordinates move by the bases BEFORE the trims (the
// reference adds sa / la counted on the untrimmed columns, :72-82), the alignment is trimmed at both ends (:68-70).
bool subhit(const Piece &hin, int start, int end, Piece &h) {
  if (end >= hin.aln.span()) end = hin.aln.span();
  if (start >= end) return false;
  h = hin;
  int sa, la, sb, lb;
  h.aln = hin.aln.slice_columns(start, end, sa, la, sb, lb);
  h.aln.trim_back();
  h.aln.trim_front();
  h.query_start += sa;
  h.query_end = h.query_start + la;
  if (h.ref->is_rc) {
    h.ref_start = h.ref_end - (lb + sb);
    h.ref_end = h.ref_end - sb;
  } else {
    h.ref_start += sb;
    h.ref_end = h.ref_start + lb;
  }
  return true;
}

// src/stats_main.cc:86-161: cut at the longest gap whose share of the alignment reaches --max-ok-gap (off by default),
// both halves again
std::vector<Piece> gap_split(const Piece &h, const StatsParams &sp) {
  struct Gap {
    int start_a, start_b, len_a, len_b;
    int start, len;
  };
  std::vector<Gap> gaps;
  Gap g{h.aln.start_a, h.aln.start_b, 0, 0, 0, 0};
  for (auto &c : h.aln.cigar) {
    if (c.second && c.first != 'M') {
      if (c.first != 'D') g.len_a = 0, g.len_b = c.second;
      else g.len_b = 0, g.len_a = c.second;
      g.len = c.second;
      gaps.push_back(g);
    }
    if (c.first != 'D') g.start_b += c.second;
    if (c.first != 'I') g.start_a += c.second;
    g.start += c.second;
  }
  // (the reference's std::sort on the same element type and comparator: equal lengths land where libstdc++ puts them)
  std::sort(gaps.begin(), gaps.end(), [](const Gap &a, const Gap &b) { return a.len > b.len; });
  std::vector<Piece> hits;
  Piece hh;
  if (sp.max_ok_gap > -1)
    for (auto &gp : gaps) {
      if (gp.start_a - h.aln.start_a < sp.min_split || gp.start_b - h.aln.start_b < sp.min_split) continue;
      if (h.aln.end_a - (gp.start_a + gp.len_a) < sp.min_split || h.aln.end_b - (gp.start_b + gp.len_b) < sp.min_split) continue;
      const double g_score = 100.0 * gp.len / (h.aln.error.matches + h.aln.error.gap_bases + h.aln.error.mismatches);
      if (g_score >= sp.max_ok_gap) {
        if (subhit(h, 0, gp.start, hh))
          for (auto &hx : gap_split(hh, sp)) hits.push_back(hx);
        if (subhit(h, gp.start + gp.len, h.aln.span(), hh))
          for (auto &hx : gap_split(hh, sp)) hits.push_back(hx);
        return hits;
      }
    }
  if (hits.empty()) hits.push_back(h);
  return hits;
}

// src/stats_main.cc:163-211: cut at runs of >= 100 N columns in either sequence, then at large gaps
std::vector<Piece> split_alignment(const Piece &h, const StatsParams &sp) {
  std::vector<Piece> hits;
  int prev_an = 0, prev_bn = 0, hit_begin = 0;
  Piece hh;
  h.aln.for_each_column([&](int i, char ca, char cb) {
    if (toupper((unsigned char)ca) == 'N') {
      prev_an++;
    } else {
      if (prev_an >= kMinAssemblyGap) {
        if (subhit(h, hit_begin, i - prev_an, hh)) hits.push_back(hh);
        hit_begin = i;
      }
      prev_an = 0;
    }
    if (toupper((unsigned char)cb) == 'N') {
      prev_bn++;
    } else {
      if (prev_bn >= kMinAssemblyGap) {
        if (subhit(h, hit_begin, i - prev_bn, hh)) hits.push_back(hh);
        hit_begin = i;
      }
      prev_bn = 0;
    }
  });
  if (!hit_begin) hits.push_back(h);
  else if (subhit(h, hit_begin, h.aln.span(), hh)) hits.push_back(hh);
  std::vector<Piece> out;
  for (auto &x : hits)
    for (auto &y : gap_split(x, sp)) out.push_back(y);
  return out;
}

struct Input {  // one BEDPE line after the swap of src/stats_main.cc:346-358
  Hit h;
  std::string cigar;
  std::string fa, fb;  // fetched (and, for fb, reverse-complemented) sequences: the pieces point into them
};
}  // namespace

long stats_generate(const std::string &ref_path, const std::string &bed_path, FILE *out, const StatsParams &sp,
                    test_cols_fn test, int device, long long *stats) {
  FastaReference fr(ref_path);
  std::ifstream fin(bed_path.c_str());
  if (!fin.is_open()) throw std::string("BED file ") + bed_path + " does not exist";
  std::vector<Input> in;
  std::string s;
  while (std::getline(fin, s)) {
    Input x;
    x.h = Hit::from_bed(s, &x.cigar);
    Hit &h = x.h;
    if (std::tie(h.query->name, h.query_start, h.query_end) > std::tie(h.ref->name, h.ref_start, h.ref_end)) {
      std::swap(h.query->name, h.ref->name);  // (names and coordinates change sides; the strands stay, :349-351)
      std::swap(h.query_start, h.ref_start);
      std::swap(h.query_end, h.ref_end);
      for (auto &c : x.cigar) c = c == 'I' ? 'D' : c == 'D' ? 'I' : c;
    }
    in.push_back(std::move(x));
  }
  // (std::sort in the reference, and the lines leave its OpenMP loop in completion order: ties and thread timing make
  // its line order arbitrary; here: stable, in sorted order)
  std::stable_sort(in.begin(), in.end(), [](const Input &a, const Input &b) {
    return std::tie(a.h.ref->is_rc, a.h.query->name, a.h.ref->name, a.h.query_start, a.h.ref_start) <
           std::tie(b.h.ref->is_rc, b.h.query->name, b.h.ref->name, b.h.query_start, b.h.ref_start);
  });

  // ---- pieces of every hit (src/stats_main.cc:213-227) ----
  std::vector<Piece> pieces;
  Params ap;
  set_alignment_scoring(ap);  // (the trims score with Globals::Align, defaults in `stats`)
  for (Input &x : in) {
    Hit &hs = x.h;
    x.fa = fr.get_sequence(hs.query->name, hs.query_start, &hs.query_end);
    x.fb = fr.get_sequence(hs.ref->name, hs.ref_start, &hs.ref_end);
    if (hs.query->is_rc) x.fa = rc(x.fa);
    if (hs.ref->is_rc) x.fb = rc(x.fb);
    if (x.cigar.empty()) throw std::string("BED line without a CIGAR in column 13");
  }
  for (Input &x : in) {  // (after the loop above: the strings do not move any more)
    Piece p{x.h.query, x.h.ref, x.h.query_start, x.h.query_end, x.h.ref_start, x.h.ref_end, Alignment(x.fa, x.fb, x.cigar)};
    for (auto &q : split_alignment(p, sp))
      if (q.aln.span() >= ap.refine_min_read) pieces.push_back(std::move(q));  // (:229: Chain::Refine::MIN_READ)
  }

  // ---- the column counters of all pieces: one device call ----
  const size_t n = pieces.size();
  std::vector<sdf_stats_cols> cols(n);
  long long columns = 0;
  {
    std::vector<sdf_stats_task> tasks(n);
    std::vector<uint32_t> runs;
    std::string pool;
    for (size_t k = 0; k < n; k++) {
      const Alignment &al = pieces[k].aln;
      sdf_stats_task &t = tasks[k];
      t.a_off = pool.size();
      t.a_len = (uint32_t)(al.end_a - al.start_a);
      pool.append(al.bases_a(), t.a_len);
      t.b_off = pool.size();
      t.b_len = (uint32_t)(al.end_b - al.start_b);
      pool.append(al.bases_b(), t.b_len);
      t.cigar_off = runs.size();
      for (auto &run : al.cigar)
        if (run.second) runs.push_back(((uint32_t)run.second << 4) | (run.first == 'M' ? 0u : run.first == 'D' ? 1u : 2u));
      t.n_cigar = (uint32_t)(runs.size() - t.cigar_off);
      t.reserved = 0;
      columns += al.span();
    }
    if (test) {
      for (size_t k = 0; k < n; k++) {
        int32_t o[16];
        test(pool.data() + tasks[k].a_off, (int)tasks[k].a_len, pool.data() + tasks[k].b_off, (int)tasks[k].b_len,
             runs.data() + tasks[k].cigar_off, (int)tasks[k].n_cigar, o, nullptr, nullptr);
        memcpy(&cols[k], o, sizeof(sdf_stats_cols));
      }
    } else if (n) {
      sdf_ctx *ctx = sdf_create(device, 0);
      if (!ctx) throw std::string("GPU backend unavailable: ") + sdf_last_error(nullptr);
      const int rc = sdf_stats_columns_batch(ctx, tasks.data(), n, pool.data(), pool.size(), runs.data(), runs.size(), cols.data());
      const std::string err = rc ? sdf_last_error(ctx) : "";
      sdf_destroy(ctx);
      if (rc) throw std::string("sdf_stats_columns_batch: ") + err;
    }
  }

  // ---- the table (src/stats_main.cc:272-336, header :371-378) ----
  fputs("#chr1\tstart1\tend1\tchr2\tstart2\tend2\tname\tscore\tstrand1\tstrand2\tmax_len\taln_len\tcomment\t"
        "indel_a\tindel_b\talnB\tmatchB\tmismatchB\ttransitionsB\ttransversions\tfracMatch\tfracMatchIndel\tjck\tk2K\t"
        "aln_gaps\tuppercaseA\tuppercaseB\tuppercaseMatches\taln_matches\taln_mismatches\taln_gaps\taln_gap_bases\t"
        "cigar\tfilter_score\n", out);
  long lines = 0;
  for (size_t k = 0; k < n; k++) {
    const Piece &h = pieces[k];
    const sdf_stats_cols &c = cols[k];
    const int align_length = h.aln.span();
    const double fracMatch = double(c.match_b) / (c.aln_b), fracMatchIndel = double(c.match_b) / (align_length);
    const double jcp = double(c.mismatch_b) / (c.aln_b), jcK = -0.75 * log(1.0 - 4.0 / 3 * jcp);
    const double p = double(c.transitions_b) / (c.aln_b), q = double(c.transversions_b) / (c.aln_b);
    const double w1 = 1.0 / (1 - 2.0 * p - q), w2 = 1.0 / (1 - 2.0 * q);
    const double k2K = 0.5 * log(w1) + 0.25 * log(w2);
    const bool same_chr = h.query->name == h.ref->name && h.query->is_rc == h.ref->is_rc;
    const int overlap = !same_chr ? 0 : std::max(0, std::min(h.query_end, h.ref_end) - std::max(h.query_start, h.ref_start));
    bool too_big_overlap = (h.query_end - h.query_start - overlap) < kBigOverlap || (h.ref_end - h.ref_start - overlap) < kBigOverlap;
    too_big_overlap &= same_chr;
    const double errorScaled = (h.aln.gaps() + h.aln.mismatches()) / double(h.aln.gaps() + h.aln.mismatches() + h.aln.matches());
    if (!(c.uppercase_a >= sp.min_uppercase && c.uppercase_b >= sp.min_uppercase && !too_big_overlap &&
          errorScaled <= sp.max_scaled_error && c.uppercase_matches >= sp.min_uppercase))
      continue;
    Hit hb;  // to_bed(false, false) of the piece with name "S" and no comment (:311-313; the reference passes &fr for its
             // translation_index, src/hit.cc:144-171, which nothing in the reference ever fills -- src/fasta.h:54 is its only
             // other mention -- so the renaming branch is dead there and has no counterpart here)
    hb.query = h.query;
    hb.ref = h.ref;
    hb.query_start = h.query_start;
    hb.query_end = h.query_end;
    hb.ref_start = h.ref_start;
    hb.ref_end = h.ref_end;
    hb.name = "S";
    hb.aln = h.aln;
    std::string line = hb.to_bed(false, false);
    auto add_i = [&](long long v) { line += "\t" + std::to_string(v); };
    auto add_d = [&](double v) { line += "\t" + format_double(v); };
    add_i(c.indel_a), add_i(c.indel_b);
    add_i(c.aln_b), add_i(c.match_b), add_i(c.mismatch_b);
    add_i(c.transitions_b), add_i(c.transversions_b);
    add_d(fracMatch), add_d(fracMatchIndel);
    add_d(jcK), add_d(k2K);
    add_i(h.aln.gaps());
    add_i(c.uppercase_a), add_i(c.uppercase_b), add_i(c.uppercase_matches);
    add_i(h.aln.matches()), add_i(h.aln.mismatches()), add_i(h.aln.gaps()), add_i(h.aln.gap_bases());
    line += "\t" + h.aln.cigar_string();
    add_d(1 - errorScaled);
    line += "\n";
    fputs(line.c_str(), out);
    ++lines;
  }
  if (stats) {
    stats[0] = (long long)in.size();
    stats[1] = (long long)n;
    stats[2] = columns;
  }
  return lines;
}

}  // namespace sdfh
