// Alignment of one candidate pair as a run-length CIGAR with a match counter (see sedef_host.h).
//
// Behaviour pinned against the reference's own class (src/align.cc, src/align.h) through oracle/_ref and the golden
// vectors in tests/golden/host_align_kat.json.gz: CIGAR strings, the four counters of populate_nice_alignment
// (src/align.cc:274-315), coordinates after trims and merges, including the reference's quirks (zero-length runs that
// count as gaps and block run merging, the dead second DP of the far-gap branch, the "\0" run of an alignment that lost
// all its columns).  The DP itself is never run here: every place the reference calls align_helper
// (src/align.cc:39-68) goes through DpSession, which records the request or replays its result.
#include <cassert>
#include <cctype>
#include <cstdio>
#include <sstream>

#include "sedef_host.h"

namespace sdfh {

// ---- character tables --------------------------------------------------------------------------------
namespace {
struct DnaTables {
  char align[128], hash[128], rev[128];
  unsigned char up[256];
  DnaTables() {
    for (int i = 0; i < 128; i++) {
      align[i] = 4;  // src/common.h:70
      hash[i] = 0;   // src/common.h:69
      rev[i] = 'N';  // src/common.h:75-77
    }
    const char *fw = "ACGT", *bw = "TGCA";
    for (int k = 0; k < 4; k++) {
      align[(int)fw[k]] = align[tolower(fw[k])] = (char)k;
      hash[(int)fw[k]] = hash[tolower(fw[k])] = (char)k;
      rev[(int)fw[k]] = bw[k];
      rev[tolower(fw[k])] = (char)tolower(bw[k]);
    }
    for (int i = 0; i < 256; i++) up[i] = (unsigned char)toupper(i);
  }
};
const DnaTables kDna;

// a match column: equal ignoring case, and not N (src/align.cc:29-35; '-' never reaches here)
inline bool same_base(char x, char y) {
  const unsigned char ux = kDna.up[(unsigned char)x], uy = kDna.up[(unsigned char)y];
  return ux == uy && ux != 'N';
}
inline double pct(double p, double tot) { return 100.0 * p / tot; }  // src/common.h:99

// which sequences a run consumes (populate_nice_alignment: D a base of a only, I a base of b only, anything else both)
inline bool takes_a(char op) { return op != 'I'; }
inline bool takes_b(char op) { return op != 'D'; }

int count_matches(const char *a, const char *b, const Cigar &cg) {
  int m = 0;
  for (auto &run : cg) {
    if (run.first == 'M')
      for (int i = 0; i < run.second; i++) m += same_base(a[i], b[i]);
    if (takes_a(run.first)) a += run.second;
    if (takes_b(run.first)) b += run.second;
  }
  return m;
}

Params g_score_params;  // Align::MATCH etc. are process-wide in the reference (src/globals.cc:25-28)
}  // namespace

void set_alignment_scoring(const Params &p) {
  // (several buckets of one process run side by side with the same scoring: no write once it is in place)
  const Params &g = g_score_params;
  if (g.match == p.match && g.mismatch == p.mismatch && g.gap_open == p.gap_open && g.gap_extend == p.gap_extend &&
      g.max_ksw_seq_len == p.max_ksw_seq_len && g.kmer == p.kmer && g.min_uppercase_match == p.min_uppercase_match &&
      g.match_chain_score == p.match_chain_score && g.max_chain_gap == p.max_chain_gap && g.refine_match == p.refine_match &&
      g.refine_mismatch == p.refine_mismatch && g.refine_gap == p.refine_gap && g.refine_gapopen == p.refine_gapopen &&
      g.refine_min_read == p.refine_min_read && g.refine_side_align == p.refine_side_align &&
      g.refine_max_gap == p.refine_max_gap && g.min_read_size == p.min_read_size && g.max_error == p.max_error)
    return;
  g_score_params = p;
}
char align_dna(char c) { return kDna.align[(unsigned char)c & 127]; }
char hash_dna(char c) { return kDna.hash[(unsigned char)c & 127]; }

std::vector<std::string> split(const std::string &s, char delim) {  // src/util.cc:33-41
  std::vector<std::string> out;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, delim)) out.push_back(item);
  return out;
}

std::string rc(const std::string &s) {  // src/util.cc:43-48
  std::string r(s.rbegin(), s.rend());
  for (auto &c : r) c = kDna.rev[(unsigned char)c & 127];
  return r;
}
void rc_inplace(char *s, size_t n) {  // the same on bases that already lie where they are used
  for (size_t i = 0, j = n; i < j; i++) {
    --j;
    const char a = kDna.rev[(unsigned char)s[i] & 127], b = kDna.rev[(unsigned char)s[j] & 127];
    s[i] = b;
    s[j] = a;
  }
}

Sequence::Sequence(const std::string &n, const std::string &s, bool is_rc_) : name(n), seq(s), is_rc(is_rc_) {
  if (is_rc) seq = rc(s);
}

// ---- DP access ---------------------------------------------------------------------------------------
Cigar DpSession::align_ranges(const char *q, int qlen, const char *t, int tlen) {
  if (recording) {
    requests->push_back(DpRequest{q, t, qlen, tlen});
    return Cigar();
  }
  assert(cursor < results->size());
  Cigar cg = (*results)[cursor++];
  // the device counts matches on the codes (ACGT + wildcard); for IUPAC letters other than N the reference's
  // comparison of the characters differs, and providers without counters leave -1: count here then
  if (cg.matches < 0 || !codes_are_exact) cg.matches = count_matches(q, t, cg);
  return cg;
}

// ---- counters ----------------------------------------------------------------------------------------
Alignment::Alignment() {}

void Alignment::recount(int matches) {  // what populate_nice_alignment derives (src/align.cc:274-315)
  matches_ = matches;
  int mcols = 0;
  error = AlignmentError{0, 0, 0, 0};
  columns_ = 0;
  for (auto &run : cigar) {
    columns_ += run.second;
    if (run.first == 'M') {
      mcols += run.second;
    } else {
      error.gaps++;  // zero-length runs count too
      error.gap_bases += run.second;
    }
  }
  error.matches = matches;
  error.mismatches = mcols - matches;
}

double Alignment::gap_error() const { return pct(error.gap_bases, error.matches + error.gap_bases + error.mismatches); }
double Alignment::mismatch_error() const {
  return pct(error.mismatches, error.matches + error.gap_bases + error.mismatches);
}

std::string Alignment::cigar_string() const {  // src/align.cc:614-621: zero-length runs are not printed
  std::string res;
  char buf[32];
  for (auto &run : cigar)
    if (run.second) {
      snprintf(buf, sizeof buf, "%d%c", run.second, run.first);
      res += buf;
    }
  return res;
}

// ---- concatenation -------------------------------------------------------------------------------------
void Alignment::append(const Cigar &piece) {  // src/align.cc:469-478
  if (piece.matches > 0) matches_ += piece.matches;
  if (piece.empty()) return;
  if (!cigar.empty() && cigar.back().first == piece.front().first) {
    cigar.back().second += piece.front().second;
    cigar.insert(cigar.end(), std::next(piece.begin()), piece.end());
  } else {
    cigar.insert(cigar.end(), piece.begin(), piece.end());
  }
}

void Alignment::prepend(const Cigar &piece) {  // src/align.cc:458-467
  if (piece.matches > 0) matches_ += piece.matches;
  if (piece.empty()) return;
  if (!cigar.empty() && cigar.front().first == piece.back().first) {
    cigar.front().second += piece.back().second;
    cigar.insert(cigar.begin(), piece.begin(), piece.begin() + (piece.size() - 1));
  } else {
    cigar.insert(cigar.begin(), piece.begin(), piece.end());
  }
}

// The stretch between two consecutive pieces (the reference has this block three times: src/align.cc:126-144,
// :233-249, :579-600).  Both sides non-empty and at most 1000 long: one DP over the stretch.  Longer: a DP over the
// first min(qgap, rgap) bases of each side and one gap run for the rest -- the reference also aligns the LAST
// min(...) bases and then compares that result's error with itself, which is never smaller, so the first variant is
// always taken and the second DP is not requested here.  When both sides are equally long that gap run has length
// zero; it stays in the run list (it counts as a gap and keeps its neighbours from merging).
void Alignment::fill_gap(int qfrom, int qgap, int rfrom, int rgap, DpSession &dp) {
  if (qgap && rgap) {
    const bool close = qgap <= 1000 && rgap <= 1000;
    const int mi = std::min(qgap, rgap);
    Cigar piece = dp.align_ranges(seq_a + qfrom, close ? qgap : mi, seq_b + rfrom, close ? rgap : mi);
    if (!close) piece.push_back({qgap == mi ? 'I' : 'D', std::max(qgap, rgap) - mi});  // (length 0 when qgap == rgap)
    append(piece);
  } else if (qgap) {
    append(Cigar{{'D', qgap}});
  } else if (rgap) {
    append(Cigar{{'I', rgap}});
  }
}

// ---- constructors ------------------------------------------------------------------------------------
Alignment::Alignment(const std::string &fa, const std::string &fb, DpSession &dp)
    : end_a((int)fa.size()), end_b((int)fb.size()), seq_a(fa.data()), seq_b(fb.data()), len_a((int)fa.size()),
      len_b((int)fb.size()) {
  cigar = dp.align_ranges(seq_a, end_a, seq_b, end_b);
  recount(std::max(cigar.matches, 0));
}

Alignment::Alignment(const std::string &fa, const std::string &fb, const std::string &cigar_str)
    : end_a((int)fa.size()), end_b((int)fb.size()), seq_a(fa.data()), seq_b(fb.data()), len_a((int)fa.size()),
      len_b((int)fb.size()) {
  int num = 0;
  for (char ch : cigar_str) {
    if (isdigit((unsigned char)ch)) num = 10 * num + (ch - '0');
    else if (ch == ';') continue;
    else {
      cigar.push_back({ch, num});
      num = 0;
    }
  }
  recount(count_matches(seq_a, seq_b, cigar));
}

Alignment::Alignment(SeqView qstr, SeqView rstr, const std::vector<Anchor> &guide, const std::vector<int> &guide_idx,
                     DpSession &dp)
    : seq_a(qstr.data()), seq_b(rstr.data()), len_a((int)qstr.size()), len_b((int)rstr.size()) {
  if (guide_idx.empty()) return;
  // anchors are exact matches (case-insensitive, no N): every column of their M runs is a match column
  const Anchor &first = guide[guide_idx[0]];
  start_a = first.q;
  start_b = first.r;
  end_a = first.q + first.l;
  end_b = first.r + first.l;
  cigar = {{'M', first.l}};
  matches_ = first.l;
  for (size_t g = 1; g < guide_idx.size(); g++) {
    const Anchor &an = guide[guide_idx[g]];
    fill_gap(end_a, an.q - end_a, end_b, an.r - end_b, dp);
    Cigar piece{{'M', an.l}};
    piece.matches = an.l;
    append(piece);
    end_a = an.q + an.l;
    end_b = an.r + an.l;
  }
  recount(matches_);
}

Alignment::Alignment(SeqView qstr, SeqView rstr, const std::vector<Hit> &guide, int side, DpSession &dp) {
  *this = guide.front().aln;
  seq_a = qstr.data();
  seq_b = rstr.data();
  len_a = (int)qstr.size();
  len_b = (int)rstr.size();
  for (size_t g = 1; g < guide.size(); g++) {
    const Hit &cur = guide[g];
    fill_gap(end_a, cur.query_start - end_a, end_b, cur.ref_start - end_b, dp);
    Cigar piece = cur.aln.cigar;
    piece.matches = cur.aln.matches_;
    append(piece);
    end_a = cur.query_end;
    end_b = cur.ref_end;
  }
  if (side) {
    // up to `side` bases on both flanks are aligned and only the best-scoring part next to the alignment is kept
    const int qlo = std::max(0, start_a - side), rlo = std::max(0, start_b - side);
    if (start_a - qlo && start_b - rlo) {
      Alignment flank;
      flank.seq_a = seq_a + qlo;
      flank.seq_b = seq_b + rlo;
      flank.end_a = start_a - qlo;
      flank.end_b = start_b - rlo;
      flank.cigar = dp.align_ranges(flank.seq_a, flank.end_a, flank.seq_b, flank.end_b);
      flank.recount(std::max(flank.cigar.matches, 0));
      flank.trim_front();
      flank.cigar.matches = flank.matches_;
      prepend(flank.cigar);
      start_a -= flank.end_a - flank.start_a;
      start_b -= flank.end_b - flank.start_b;
    }
    const int qhi = std::min(end_a + side, len_a), rhi = std::min(end_b + side, len_b);
    if (qhi - end_a && rhi - end_b) {
      Alignment flank;
      flank.seq_a = seq_a + end_a;
      flank.seq_b = seq_b + end_b;
      flank.end_a = qhi - end_a;
      flank.end_b = rhi - end_b;
      flank.cigar = dp.align_ranges(flank.seq_a, flank.end_a, flank.seq_b, flank.end_b);
      flank.recount(std::max(flank.cigar.matches, 0));
      flank.trim_back();
      flank.cigar.matches = flank.matches_;
      append(flank.cigar);
      end_a += flank.end_a;
      end_b += flank.end_b;
    }
  }
  recount(matches_);
}

// ---- trims -------------------------------------------------------------------------------------------
namespace {
// Walks the columns of a run list in either direction and scores them like the reference's trim scans: a match
// column scores `match`, a mismatch column `mismatch`, a gap column `gap_extend`, plus `gap_open` when it is the first
// column of the scan or its neighbour towards the scan's origin is not a gap in the same sequence
// (src/align.cc:347-362, :404-418).
struct ColumnScan {
  const Params &p;
  int prev_kind = -1;  // 0: both bases, 1: gap in a, 2: gap in b
  explicit ColumnScan(const Params &pp) : p(pp) {}
  int gap(int kind) {
    const int s = (prev_kind != kind ? p.gap_open : 0) + p.gap_extend;
    prev_kind = kind;
    return s;
  }
  int pair(bool match) {
    prev_kind = 0;
    return match ? p.match : p.mismatch;
  }
};
}  // namespace

void Alignment::trim_front() {
  const Params &p = g_score_params;
  ColumnScan scan(p);
  // best suffix: scan from the last column; ties go to the longer suffix (>=)
  const int none = end_a - start_a;  // the reference's "nothing found" marker is the LENGTH OF a, not of the columns
  int best = 0, best_col = none, best_matches = 0, score = 0, seen = 0;
  int ia = end_a, ib = end_b, col = columns_;
  for (size_t k = cigar.size(); k-- > 0;) {
    const char op = cigar[k].first;
    for (int i = 0; i < cigar[k].second; i++) {
      --col;
      if (op == 'M') {
        const bool m = same_base(seq_a[--ia], seq_b[--ib]);
        seen += m;
        score += scan.pair(m);
      } else if (!takes_a(op)) {
        --ib;
        score += scan.gap(1);
      } else if (!takes_b(op)) {
        --ia;
        score += scan.gap(2);
      } else {  // (a run that consumes both and is not M: scored as a mismatch column, like the reference)
        --ia;
        --ib;
        score += scan.pair(false);
      }
      if (score >= best) best = score, best_col = col, best_matches = seen;
    }
  }
  if (best_col == none) {
    start_a = end_a;
    start_b = end_b;
    cigar.clear();
    recount(0);
    return;
  }
  // drop the first best_col columns (the cut lands inside an M run: a best suffix starts with a match)
  int done = 0;
  for (size_t k = 0; k < cigar.size(); k++) {
    const int len = cigar[k].second;
    if (done + len > best_col) {
      const int need = best_col - done;
      cigar[k].second -= need;
      for (size_t j = 0; j < k; j++) cigar.pop_front();
      start_a += need;
      start_b += need;
      break;
    }
    done += len;
    if (takes_a(cigar[k].first)) start_a += len;
    if (takes_b(cigar[k].first)) start_b += len;
  }
  recount(best_matches);
}

void Alignment::trim_back() {
  const Params &p = g_score_params;
  ColumnScan scan(p);
  int best = 0, best_col = -1, best_matches = 0, score = 0, seen = 0;
  int ia = start_a, ib = start_b, col = 0;
  for (size_t k = 0; k < cigar.size(); k++) {
    const char op = cigar[k].first;
    for (int i = 0; i < cigar[k].second; i++, col++) {
      if (op == 'M') {
        const bool m = same_base(seq_a[ia++], seq_b[ib++]);
        seen += m;
        score += scan.pair(m);
      } else if (!takes_a(op)) {
        ++ib;
        score += scan.gap(1);
      } else if (!takes_b(op)) {
        ++ia;
        score += scan.gap(2);
      } else {
        ++ia;
        ++ib;
        score += scan.pair(false);
      }
      if (score >= best) best = score, best_col = col, best_matches = seen;
    }
  }
  if (best_col == -1) {
    end_a = start_a;
    end_b = start_b;
    cigar.clear();
    recount(0);
    return;
  }
  const int keep = best_col + 1;  // columns kept
  end_a = start_a;
  end_b = start_b;
  int done = 0;
  for (size_t k = 0; k < cigar.size(); k++) {
    const int len = cigar[k].second;
    if (done + len >= keep) {
      const int need = keep - done;
      cigar[k].second = need;
      while (cigar.size() - 1 > k) cigar.pop_back();
      end_a += need;
      end_b += need;
      break;
    }
    done += len;
    if (takes_a(cigar[k].first)) end_a += len;
    if (takes_b(cigar[k].first)) end_b += len;
  }
  recount(best_matches);
}

// ---- a range of columns (reference: subhit, src/stats_main.cc:33-84, up to the trims) ------------------------
Alignment Alignment::slice_columns(int start, int end, int &sa, int &la, int &sb, int &lb) const {
  Alignment out;
  out.seq_a = seq_a;
  out.seq_b = seq_b;
  out.len_a = len_a;
  out.len_b = len_b;
  sa = la = sb = lb = 0;
  int col = 0;
  for (auto &run : cigar) {
    const int lo = std::max(col, start), hi = std::min(col + run.second, end);
    const int before = std::min(col + run.second, start) - col;  // columns of the run before `start`
    if (before > 0) {
      if (takes_a(run.first)) sa += before;
      if (takes_b(run.first)) sb += before;
    }
    if (hi > lo) {
      out.cigar.push_back({run.first, hi - lo});
      if (takes_a(run.first)) la += hi - lo;
      if (takes_b(run.first)) lb += hi - lo;
    }
    col += run.second;
    if (col >= end) break;
  }
  out.start_a = start_a + sa;
  out.end_a = out.start_a + la;
  out.start_b = start_b + sb;
  out.end_b = out.start_b + lb;
  out.normalise();
  out.recount(count_matches(seq_a + out.start_a, seq_b + out.start_b, out.cigar));
  return out;
}

// ---- merge -------------------------------------------------------------------------------------------
// Columns leave from the end until `trim` bases of the query (or of the reference) have gone with them; a column that
// does not consume that sequence leaves too while the count is still short (src/align.cc:511-517, :543-549).
// Returns nothing; coordinates and the match counter follow the columns.
int Alignment::cut_tail(int trim, bool by_query) {
  int gone = 0;
  while (gone < trim && !cigar.empty()) {
    auto &run = cigar.back();
    const bool counts = by_query ? takes_a(run.first) : takes_b(run.first);
    const int take = counts ? std::min(run.second, trim - gone) : run.second;
    if (run.first == 'M')
      for (int i = 1; i <= take; i++) matches_ -= same_base(seq_a[end_a - i], seq_b[end_b - i]);
    if (takes_a(run.first)) end_a -= take;
    if (takes_b(run.first)) end_b -= take;
    if (counts) gone += take;
    run.second -= take;
    if (run.second == 0) cigar.pop_back();
  }
  return gone;
}

int Alignment::cut_head(int trim, bool by_query) {  // (src/align.cc:524-534, :556-566)
  int gone = 0;
  while (gone < trim && !cigar.empty()) {
    auto &run = cigar.front();
    const bool counts = by_query ? takes_a(run.first) : takes_b(run.first);
    const int take = counts ? std::min(run.second, trim - gone) : run.second;
    if (run.first == 'M')
      for (int i = 0; i < take; i++) matches_ -= same_base(seq_a[start_a + i], seq_b[start_b + i]);
    if (takes_a(run.first)) start_a += take;
    if (takes_b(run.first)) start_b += take;
    if (counts) gone += take;
    run.second -= take;
    if (run.second == 0) cigar.pop_front();
  }
  return gone;
}

// The reference rebuilds the CIGAR from the remaining columns (cigar_from_alignment, src/align.cc:480-501): runs of
// length zero disappear, neighbours of one kind fuse, and an alignment without columns gets one run {'\0', 0}.
void Alignment::normalise() {
  Cigar out;
  for (auto &run : cigar) {
    if (run.second == 0) continue;
    if (!out.empty() && out.back().first == run.first) out.back().second += run.second;
    else out.push_back(run);
  }
  if (out.empty()) out.push_back({'\0', 0});
  cigar = out;
}

void Alignment::merge(Alignment &cur, SeqView qstr, SeqView rstr, DpSession &dp) {
  seq_a = cur.seq_a = qstr.data();
  seq_b = cur.seq_b = rstr.data();
  len_a = cur.len_a = (int)qstr.size();
  len_b = cur.len_b = (int)rstr.size();
  // the overlap goes from BOTH alignments: first what overlaps in the query, then what still overlaps in the reference
  for (int pass = 0; pass < 2; pass++) {
    const bool by_query = pass == 0;
    const int trim = by_query ? end_a - cur.start_a : end_b - cur.start_b;
    cut_tail(trim, by_query);
    cur.cut_head(trim, by_query);
  }
  normalise();
  cur.normalise();
  fill_gap(end_a, cur.start_a - end_a, end_b, cur.start_b - end_b, dp);
  Cigar piece = cur.cigar;
  piece.matches = cur.matches_;
  append(piece);
  end_a = cur.end_a;
  end_b = cur.end_b;
  recount(matches_);
}

}  // namespace sdfh
