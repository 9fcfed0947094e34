// Alignment object of the `sedef align` stage (restates reference src/align.cc, src/align.h).
// The DP itself is never run here: every place the reference calls align_helper
// (src/align.cc:39-68) goes through DpSession::align, which records or replays.
#include <algorithm>
#include <cassert>
#include <cctype>
#include <cstdio>
#include <sstream>

#include "sedef_host.h"

namespace sdfh {

// ---- small utilities -------------------------------------------------------------------------------
namespace {
struct DnaTables {
  char align[128], hash[128], rev[128];
  DnaTables() {
    for (int i = 0; i < 128; i++) {
      align[i] = 4;   // src/common.h:70
      hash[i] = 0;    // src/common.h:69
      rev[i] = 'N';   // src/common.h:75-77
    }
    const char *fw = "ACGT", *bw = "TGCA";
    for (int k = 0; k < 4; k++) {
      align[(int)fw[k]] = align[tolower(fw[k])] = (char)k;
      hash[(int)fw[k]] = hash[tolower(fw[k])] = (char)k;
      rev[(int)fw[k]] = bw[k];
      rev[tolower(fw[k])] = (char)tolower(bw[k]);
    }
  }
};
const DnaTables kDna;

struct UpperTable {  // toupper() of every byte, asked once (ceq runs once per alignment column)
  unsigned char up[256];
  UpperTable() {
    for (int i = 0; i < 256; i++) up[i] = (unsigned char)toupper(i);
  }
};
const UpperTable kUpper;

inline bool ceq(char x, char y) {  // src/align.cc:29-35
  if (x == '-' || y == '-') return false;
  const unsigned char ux = kUpper.up[(unsigned char)x], uy = kUpper.up[(unsigned char)y];
  if (ux == 'N' || uy == 'N') return false;
  return ux == uy;
}
inline double pct(double p, double tot) { return 100.0 * p / tot; }  // src/common.h:99
}  // namespace

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

Sequence::Sequence(const std::string &n, const std::string &s, bool is_rc_) : name(n), seq(s), is_rc(is_rc_) {
  if (is_rc) seq = rc(s);
}

Cigar DpSession::align(const std::string &q_codes, const std::string &t_codes) {
  if (recording) {
    requests->push_back({q_codes, t_codes});
    return Cigar();
  }
  assert(cursor < results->size());
  return (*results)[cursor++];
}

// ---- constructors ------------------------------------------------------------------------------------
Alignment::Alignment() {}

Alignment::Alignment(const std::string &fa, const std::string &fb, DpSession &dp)  // src/align.cc:76-88
    : chr_a("A"), start_a(0), end_a((int)fa.size()), chr_b("B"), start_b(0), end_b((int)fb.size()), a(fa), b(fb) {
  std::string xa = fa, xb = fb;
  for (auto &c : xa) c = align_dna(c);
  for (auto &c : xb) c = align_dna(c);
  cigar = dp.align(xa, xb);
  populate_nice_alignment();
}

Alignment::Alignment(const std::string &fa, const std::string &fb, const std::string &cigar_str)  // :90-105
    : chr_a("A"), start_a(0), end_a((int)fa.size()), chr_b("B"), start_b(0), end_b((int)fb.size()), a(fa), b(fb) {
  int num = 0;
  for (char ch : cigar_str) {
    if (isdigit((unsigned char)ch)) num = 10 * num + (ch - '0');
    else if (ch == ';') continue;
    else {
      cigar.push_back({ch, num});
      num = 0;
    }
  }
  populate_nice_alignment();
}

namespace {
// the gap between two consecutive pieces (identical in the three places the reference has it:
// src/align.cc:126-144, :233-249, :579-600)
void fill_gap(Alignment &self, const std::string &qstr, const std::string &rstr, int qfrom, int qgap, int rfrom,
              int rgap, int q_next, int r_next, DpSession &dp) {
  if (qgap && rgap) {
    if (qgap <= 1000 && rgap <= 1000) {  // "close" pieces: one DP over the whole gap
      Alignment gap(qstr.substr(qfrom, qgap), rstr.substr(rfrom, rgap), dp);
      self.append_cigar(gap.cigar);
    } else {  // far: DP over the first min(qgap,rgap) bases, the rest is one gap run
      const int ma = std::max(qgap, rgap), mi = std::min(qgap, rgap);
      Alignment ma1(qstr.substr(qfrom, mi), rstr.substr(rfrom, mi), dp);
      ma1.cigar.push_back({qgap == mi ? 'I' : 'D', ma - mi});
      // The reference also aligns the LAST mi bases (ma2) and then compares
      // ma2.total_error() < ma2.total_error(), which is never true: ma1 is always taken and ma2's DP
      // has no observable effect, so it is not requested.
      (void)q_next;
      (void)r_next;
      self.append_cigar(ma1.cigar);
    }
  } else if (qgap) {
    self.append_cigar({{'D', qgap}});
  } else if (rgap) {
    self.append_cigar({{'I', rgap}});
  }
}
}  // namespace

Alignment::Alignment(const std::string &qstr, const std::string &rstr, const std::vector<Hit> &guide, int side,
                     DpSession &dp) {  // src/align.cc:107-197
  auto prev = guide.begin();
  *this = prev->aln;
  for (auto cur = std::next(prev); cur != guide.end(); ++cur) {
    const int qs = cur->query_start, qe = cur->query_end, qpe = prev->query_end;
    const int rs = cur->ref_start, re = cur->ref_end, rpe = prev->ref_end;
    end_a = qe;
    end_b = re;
    a += qstr.substr(qpe, qe - qpe);
    b += rstr.substr(rpe, re - rpe);
    fill_gap(*this, qstr, rstr, qpe, qs - qpe, rpe, rs - rpe, qs, rs, dp);
    append_cigar(cur->aln.cigar);
    prev = cur;
  }
  int qlo = start_a, qhi = end_a, rlo = start_b, rhi = end_b;
  if (side) {
    int qlo_n = std::max(0, qlo - side), rlo_n = std::max(0, rlo - side);
    if (qlo - qlo_n && rlo - rlo_n) {
      Alignment gap(qstr.substr(qlo_n, qlo - qlo_n), rstr.substr(rlo_n, rlo - rlo_n), dp);
      gap.trim_front();
      qlo_n = qlo - (gap.end_a - gap.start_a);
      rlo_n = rlo - (gap.end_b - gap.start_b);
      prepend_cigar(gap.cigar);
      a = qstr.substr(qlo_n, qlo - qlo_n) + a;
      b = rstr.substr(rlo_n, rlo - rlo_n) + b;
      start_a = qlo = qlo_n;
      start_b = rlo = rlo_n;
    }
    int qhi_n = std::min(qhi + side, (int)qstr.size()), rhi_n = std::min(rhi + side, (int)rstr.size());
    if (qhi_n - qhi && rhi_n - rhi) {
      Alignment gap(qstr.substr(qhi, qhi_n - qhi), rstr.substr(rhi, rhi_n - rhi), dp);
      gap.trim_back();
      qhi_n = qhi + gap.end_a;
      rhi_n = rhi + gap.end_b;
      append_cigar(gap.cigar);
      a += qstr.substr(qhi, qhi_n - qhi);
      b += rstr.substr(rhi, rhi_n - rhi);
      end_a = qhi = qhi_n;
      end_b = rhi = rhi_n;
    }
  }
  populate_nice_alignment();
}

Alignment::Alignment(const std::string &qstr, const std::string &rstr, const std::vector<Anchor> &guide,
                     const std::vector<int> &guide_idx, DpSession &dp)  // src/align.cc:199-270
    : chr_a("A"), chr_b("B") {
  if (guide_idx.empty()) {
    *this = Alignment();
    return;
  }
  auto prev = guide_idx.begin();
  start_a = guide[*prev].q;
  end_a = guide[*prev].q + guide[*prev].l;
  start_b = guide[*prev].r;
  end_b = guide[*prev].r + guide[*prev].l;
  a = qstr.substr(start_a, end_a - start_a);
  b = rstr.substr(start_b, end_b - start_b);
  cigar = {{'M', end_a - start_a}};
  for (auto cur = std::next(prev); cur != guide_idx.end(); ++cur) {
    const int qs = guide[*cur].q, qe = qs + guide[*cur].l, qpe = guide[*prev].q + guide[*prev].l;
    const int rs = guide[*cur].r, re = rs + guide[*cur].l, rpe = guide[*prev].r + guide[*prev].l;
    end_a = qe;
    end_b = re;
    a += qstr.substr(qpe, qe - qpe);
    b += rstr.substr(rpe, re - rpe);
    fill_gap(*this, qstr, rstr, qpe, qs - qpe, rpe, rs - rpe, qs, rs, dp);
    append_cigar({{'M', qe - qs}});
    prev = cur;
  }
  populate_nice_alignment();
}

// ---- column strings and counters (src/align.cc:274-315) -------------------------------------------------
void Alignment::populate_nice_alignment() {
  // (one pass over preallocated strings; the reference appends column by column and counts in a second pass --
  // same strings, same counters)
  size_t cols = 0;
  for (auto &c : cigar) cols += c.second > 0 ? (size_t)c.second : 0;
  align_a.resize(cols);
  align_b.resize(cols);
  alignment.resize(cols);
  char *pa = &align_a[0], *pb = &align_b[0], *pm = &alignment[0];
  const char *sa = a.data(), *sb = b.data();
  size_t ia = 0, ib = 0, o = 0;
  error = AlignmentError{0, 0, 0, 0};
  for (auto &c : cigar) {
    const int n = c.second;
    if (c.first == 'M') {
      for (int i = 0; i < n; i++, o++) {
        const char ca = sa[ia++], cb = sb[ib++];
        const bool eq = ceq(ca, cb);
        pm[o] = eq ? '|' : '*';
        pa[o] = ca;
        pb[o] = cb;
        if (ca != '-' && cb != '-') {
          if (eq) error.matches++; else error.mismatches++;
        }
      }
    } else {
      error.gaps++;  // zero-length runs count too
      error.gap_bases += n;
      // as in the reference: op D takes a base of a only, op I a base of b only, any other op one of each
      const bool take_a = c.first != 'I', take_b = c.first != 'D';
      for (int i = 0; i < n; i++, o++) {
        const char ca = take_a ? sa[ia++] : '-', cb = take_b ? sb[ib++] : '-';
        pm[o] = '*';
        pa[o] = ca;
        pb[o] = cb;
        if (ca != '-' && cb != '-') {
          if (ceq(ca, cb)) error.matches++; else error.mismatches++;
        }
      }
    }
  }
}

double Alignment::gap_error() const { return pct(error.gap_bases, error.matches + error.gap_bases + error.mismatches); }
double Alignment::mismatch_error() const {
  return pct(error.mismatches, error.matches + error.gap_bases + error.mismatches);
}

void Alignment::trim() {  // src/align.cc:317-341
  while (!cigar.empty()) {
    if (cigar[0].first == 'D') {
      a = a.substr(cigar[0].second);
      start_a += cigar[0].second;
      cigar.pop_front();
    } else if (cigar[0].first == 'I') {
      b = b.substr(cigar[0].second);
      start_b += cigar[0].second;
      cigar.pop_front();
    } else if (cigar.back().first == 'D') {
      end_a -= cigar.back().second;
      a = a.substr(0, a.size() - cigar.back().second);
      cigar.pop_back();
    } else if (cigar.back().first == 'I') {
      end_b -= cigar.back().second;
      b = b.substr(0, b.size() - cigar.back().second);
      cigar.pop_back();
    } else {
      break;
    }
  }
  populate_nice_alignment();
}

namespace {
// score contribution of alignment column i given its neighbour towards the scan origin
inline int column_score(const Alignment &al, int i, bool first, int nb, const Params &p) {
  if (al.alignment[i] == '|') return p.match;
  if (al.align_a[i] != '-' && al.align_b[i] != '-') return p.mismatch;
  int s = 0;
  if (first || (al.align_a[i] == '-' && al.align_a[nb] != '-') || (al.align_b[i] == '-' && al.align_b[nb] != '-'))
    s += p.gap_open;
  return s + p.gap_extend;
}
Params g_score_params;  // Align::MATCH etc. are process-wide in the reference (src/globals.cc:25-28)
}  // namespace

void set_alignment_scoring(const Params &p) { g_score_params = p; }

void Alignment::trim_front() {  // ABCD -> --CD  (src/align.cc:343-398)
  const Params &p = g_score_params;
  int max_score = 0, max_i = (int)a.size(), score = 0;
  const int n = (int)alignment.size();
  for (int i = n - 1; i >= 0; i--) {
    score += column_score(*this, i, i == n - 1, i + 1 < n ? i + 1 : i, p);
    if (score >= max_score) max_score = score, max_i = i;
  }
  if (max_i == (int)a.size()) {
    a = "";
    b = "";
    start_a = end_a;
    start_b = end_b;
    cigar.clear();
    return;
  }
  for (int ci = 0, cur_len = 0; ci < (int)cigar.size(); ci++) {
    if (cigar[ci].second + cur_len > max_i) {
      const int need = max_i - cur_len;
      cigar[ci].second -= need;
      for (int cj = 0; cj < ci; cj++) cigar.pop_front();
      start_a += need;
      start_b += need;
      break;
    }
    cur_len += cigar[ci].second;
    if (cigar[ci].first == 'M') {
      start_a += cigar[ci].second;
      start_b += cigar[ci].second;
    } else if (cigar[ci].first == 'I') {
      start_b += cigar[ci].second;
    } else {
      start_a += cigar[ci].second;
    }
  }
  a = a.substr(start_a, end_a - start_a);
  b = b.substr(start_b, end_b - start_b);
  populate_nice_alignment();
}

void Alignment::trim_back() {  // ABCD -> AB--  (src/align.cc:400-456)
  const Params &p = g_score_params;
  int max_score = 0, max_i = -1, score = 0;
  const int n = (int)alignment.size();
  for (int i = 0; i < n; i++) {
    score += column_score(*this, i, i == 0, i > 0 ? i - 1 : i, p);
    if (score >= max_score) max_score = score, max_i = i;
  }
  if (max_i == -1) {
    a = "";
    b = "";
    end_a = start_a;
    end_b = start_b;
    cigar.clear();
    return;
  }
  max_i++;
  end_a = start_a, end_b = start_b;
  for (int ci = 0, cur_len = 0; ci < (int)cigar.size(); ci++) {
    if (cigar[ci].second + cur_len >= max_i) {
      const int need = max_i - cur_len;
      cigar[ci].second = need;
      while ((int)cigar.size() - 1 > ci) cigar.pop_back();
      end_a += need;
      end_b += need;
      break;
    }
    cur_len += cigar[ci].second;
    if (cigar[ci].first == 'M') {
      end_a += cigar[ci].second;
      end_b += cigar[ci].second;
    } else if (cigar[ci].first == 'I') {
      end_b += cigar[ci].second;
    } else {
      end_a += cigar[ci].second;
    }
  }
  a = a.substr(start_a, end_a - start_a);
  b = b.substr(start_b, end_b - start_b);
  populate_nice_alignment();
}

void Alignment::prepend_cigar(const Cigar &app) {  // src/align.cc:458-467
  if (app.empty()) return;
  if (!cigar.empty() && cigar.front().first == app.back().first) {
    cigar.front().second += app.back().second;
    cigar.insert(cigar.begin(), app.begin(), app.begin() + (app.size() - 1));
  } else {
    cigar.insert(cigar.begin(), app.begin(), app.end());
  }
}

void Alignment::append_cigar(const Cigar &app) {  // src/align.cc:469-478
  if (app.empty()) return;
  if (!cigar.empty() && cigar.back().first == app.front().first) {
    cigar.back().second += app.front().second;
    cigar.insert(cigar.end(), std::next(app.begin()), app.end());
  } else {
    cigar.insert(cigar.end(), app.begin(), app.end());
  }
}

void Alignment::cigar_from_alignment() {  // src/align.cc:480-501
  cigar.clear();
  int sz = 0;
  char op = 0, top;
  for (size_t i = 0; i < alignment.size(); i++) {
    if (align_a[i] == '-') top = 'I';
    else if (align_b[i] == '-') top = 'D';
    else top = 'M';
    if (op != top) {
      if (op) cigar.push_back({op, sz});
      op = top, sz = 0;
    }
    sz++;
  }
  cigar.push_back({op, sz});
}

void Alignment::merge(Alignment &cur, const std::string &qstr, const std::string &rstr, DpSession &dp) {
  // src/align.cc:505-610: cut the overlapping columns off both alignments, first by query then by reference
  for (int pass = 0; pass < 2; pass++) {
    const int trim = pass == 0 ? end_a - cur.start_a : end_b - cur.start_b;
    int q = 0, r = 0, i;
    for (i = (int)alignment.size() - 1; i >= 0 && (pass == 0 ? q : r) < trim; i--) {
      if (align_a[i] != '-') q++;
      if (align_b[i] != '-') r++;
    }
    align_a = align_a.substr(0, i + 1);
    alignment = alignment.substr(0, i + 1);
    align_b = align_b.substr(0, i + 1);
    end_a = start_a + (int)a.size() - q;
    end_b = start_b + (int)b.size() - r;
    a = a.substr(0, a.size() - q);
    b = b.substr(0, b.size() - r);

    q = 0, r = 0;
    for (i = 0; i < (int)cur.alignment.size() && (pass == 0 ? q : r) < trim; i++) {
      if (cur.align_a[i] != '-') q++;
      if (cur.align_b[i] != '-') r++;
    }
    cur.align_a = cur.align_a.substr(i);
    cur.alignment = cur.alignment.substr(i);
    cur.align_b = cur.align_b.substr(i);
    cur.start_a += q;
    cur.start_b += r;
    cur.a = cur.a.substr(q);
    cur.b = cur.b.substr(r);
  }
  cigar_from_alignment();
  cur.cigar_from_alignment();

  const int qgap = cur.start_a - end_a, rgap = cur.start_b - end_b;
  fill_gap(*this, qstr, rstr, end_a, qgap, end_b, rgap, cur.start_a, cur.start_b, dp);
  a += qstr.substr(end_a, qgap) + cur.a;
  b += rstr.substr(end_b, rgap) + cur.b;
  end_a = cur.end_a;
  end_b = cur.end_b;
  append_cigar(cur.cigar);
  populate_nice_alignment();
}

std::string Alignment::cigar_string() const {  // src/align.cc:614-621
  std::string res;
  char buf[32];
  for (auto &p : cigar)
    if (p.second) {
      snprintf(buf, sizeof buf, "%d%c", p.second, p.first);
      res += buf;
    }
  return res;
}

void Alignment::swap() {  // src/align.cc:623-636
  std::swap(a, b);
  std::swap(chr_a, chr_b);
  std::swap(start_a, start_b);
  std::swap(end_a, end_b);
  for (auto &p : cigar)
    if (p.second) {
      if (p.first == 'I') p.first = 'D';
      else if (p.first == 'D') p.first = 'I';
    }
  populate_nice_alignment();
}

}  // namespace sdfh
