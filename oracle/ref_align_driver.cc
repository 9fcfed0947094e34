// TEST INFRASTRUCTURE: thin C driver around the REFERENCE's own Alignment / Hit classes
// (src/align.{h,cc}, src/hit.{h,cc}), compiled together with the reference sources that build
// without Boost (see Makefile `refalign`).  It only constructs objects through the reference's public
// interface and prints what it returns; the refinement loop below calls the reference's merge() and
// guide constructor in the order refine_chains does (src/refine.cc:163-183).
#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "align.h"
#include "fasta.h"
#include "globals.h"
#include "hash.h"
#include "hit.h"
#include "merge.h"
#include "segment.h"

using namespace std;

static vector<vector<Anchor>> parse_chains(const char *spec) {
  vector<vector<Anchor>> chains(1);
  int q, r, l, n = 0;
  const char *p = spec;
  while (*p) {
    if (*p == '|') {
      chains.push_back({});
      p++;
      continue;
    }
    if (sscanf(p, "%d,%d,%d%n", &q, &r, &l, &n) == 3) {
      chains.back().push_back(Anchor{q, r, l, 0});
      p += n;
      if (*p == ';') p++;
    } else {
      break;
    }
  }
  if (chains.back().empty()) chains.pop_back();
  return chains;
}

static int emit(const string &s, char *buf, size_t cap) {
  if (s.size() + 1 > cap) return -2;
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

extern "C" {

int ref_alignment_pair(const char *fa, const char *fb, char *cigar, size_t cap, int *counts) {
  Alignment al{string(fa), string(fb)};
  counts[0] = al.matches();
  counts[1] = al.mismatches();
  counts[2] = al.gaps();
  counts[3] = al.gap_bases();
  counts[4] = al.span();
  return emit(al.cigar_string(), cigar, cap);
}

// chains: "q,r,l;q,r,l|q,r,l;..." (anchors of each chain in order).  Builds every chain alignment
// (src/align.cc:199-270), merges overlapping neighbours and builds the guide alignment with side
// extension.  Output: "qs qe rs re cigar matches mismatches gaps gap_bases|to_bed line".
int ref_guide_from_chains(const char *qstr_, const char *rstr_, const char *spec, int side, char *out, size_t cap) {
  const string qstr = qstr_, rstr = rstr_;
  auto chains = parse_chains(spec);
  auto qs = make_shared<Sequence>("QRY", qstr);
  auto rs = make_shared<Sequence>("REF", rstr);
  vector<Hit> hits;
  for (auto &c : chains) {
    vector<int> idx;
    for (int i = 0; i < (int)c.size(); i++) idx.push_back(i);
    Hit h{qs, 0, 0, rs, 0, 0, 0, "", "", {}};
    h.aln = Alignment(qstr, rstr, c, idx);
    update_from_alignment(h);
    hits.push_back(h);
  }
  vector<Hit> guide;
  Hit *prev = &hits[0];
  for (int pi = 1; pi < (int)hits.size(); pi++) {
    auto &cur = hits[pi];
    if (cur.query_start < prev->query_end || cur.ref_start < prev->ref_end) {
      prev->aln.merge(cur.aln, qstr, rstr);
      update_from_alignment(*prev);
    } else {
      guide.push_back(*prev);
      prev = &cur;
    }
  }
  guide.push_back(*prev);
  Hit fin{qs, 0, 0, rs, 0, 0, 0, "name", "cmt", {}};
  fin.aln = Alignment(qstr, rstr, guide, side);
  update_from_alignment(fin);
  ostringstream os;
  os << fin.query_start << " " << fin.query_end << " " << fin.ref_start << " " << fin.ref_end << " "
     << fin.aln.cigar_string() << " " << fin.aln.matches() << " " << fin.aln.mismatches() << " " << fin.aln.gaps()
     << " " << fin.aln.gap_bases() << "|" << fin.to_bed(false);
  return emit(os.str(), out, cap);
}

// merge() (src/merge.cc:35-109).  Hits arrive as "qname qs qe rname rs re rc" per line; Hit objects are
// filled field by field (Hit::from_bed needs split(), which lives in the Boost-dependent src/util.cc).
int ref_merge(const char *spec, int merge_dist, char *out, size_t cap) {
  vector<Hit> hits;
  istringstream is(spec);
  string qn, rn;
  int qs, qe, rs, re, rcf;
  while (is >> qn >> qs >> qe >> rn >> rs >> re >> rcf) {
    auto q = make_shared<Sequence>(qn, "", false);
    auto r = make_shared<Sequence>(rn, "", false);
    r->is_rc = rcf != 0;
    hits.push_back(Hit{q, qs, qe, r, rs, re, 0, "", "", {}});
  }
  auto res = merge(hits, merge_dist);
  ostringstream os;
  for (auto &h : res) os << h.to_bed(false) << "\n";
  return emit(os.str(), out, cap);
}

// FastaReference::get_sequence (src/fasta.cc:105-142).  The reference parses the .fai with split() (src/util.cc, needs
// Boost.Math, not compiled): `fasta_path` must have NO .fai next to it, so the constructor leaves the index empty, and
// the one index entry is inserted here through the public FastaIndex map.  `has_end` = 0 passes end = nullptr.
int ref_fasta_get(const char *fasta_path, const char *name, int length, long long offset, int line_blen, int line_len,
                  int start, int has_end, int *end, char *out, size_t cap) {
  try {
    FastaReference fr{string(fasta_path)};
    fr.index.insert(make_pair(string(name), FastaIndexEntry(name, length, offset, line_blen, line_len)));
    string s = fr.get_sequence(name, start, has_end ? end : nullptr);
    return emit(s, out, cap);
  } catch (string &e) {
    emit(e, out, cap);
    return -1;
  }
}

// Hit::extend (src/hit.cc:200-207): io = {query_start, query_end, ref_start, ref_end}
int ref_hit_extend(int *io, double factor, int max_extend) {
  auto q = make_shared<Sequence>("q", "", false);
  auto r = make_shared<Sequence>("r", "", false);
  Hit h{q, io[0], io[1], r, io[2], io[3], 0, "", "", {}};
  h.extend(factor, max_extend);
  io[0] = h.query_start;
  io[1] = h.query_end;
  io[2] = h.ref_start;
  io[3] = h.ref_end;
  return 0;
}

// Sequence ctor (src/hash.cc:104-109) with is_rc = false (is_rc = true calls rc(), src/util.cc, not compiled).
// Output: "name|seq|is_rc".
int ref_sequence(const char *name, const char *seq, char *out, size_t cap) {
  Sequence s{string(name), string(seq), false};
  return emit(s.name + "|" + s.seq + "|" + (s.is_rc ? "1" : "0"), out, cap);
}

// Alignment(fa, fb, cigar) (src/align.cc:90-106): the column strings populate_nice_alignment expands and its
// AlignmentError counters -- what `stats generate` walks (src/stats_main.cc:222-270).  The strings are private; the
// reference's own Alignment::print(-1, true) returns them (src/align.cc:658-661).  out: "align_a\nalignment\nalign_b\n\n".
int ref_alignment_from_cigar(const char *fa, const char *fb, const char *cigar, char *out, size_t cap, int *counts) {
  Alignment al{string(fa), string(fb), string(cigar)};
  counts[0] = al.matches();
  counts[1] = al.mismatches();
  counts[2] = al.gaps();
  counts[3] = al.gap_bases();
  counts[4] = al.span();
  return emit(al.span() ? al.print(-1, true) : string("\n\n\n\n"), out, cap);
}

// SegmentTree<T> (src/segment.h:21-56, src/segment.tpp:12-172), the priority search tree chain_anchors keeps its
// candidate predecessors in, instantiated on a point type shaped like chain_anchors' own (`Coor`, src/chain.cc:106-110:
// that struct is local to a function of a file that cannot be compiled here) and driven through its public interface
// by a script: which of several equally good points a range query returns is decided by the tree's shape and history
// (the `>=` / `>` of segment.tpp:62,89,128), and this is where that rule is pinned.
//   pts[2 i], pts[2 i + 1]: x of point i (pos = i); the constructor sorts them (n >= 2: it takes clz(n - 1)).
//   ops[5 k ..]: {0, x.first, x.second, score, -} activate; {1, x.first, x.second, -, -} deactivate;
//                {2, p.first, p.second, q.first, q.second} rmq(p, q) -> out[2 k] = pos of the returned point (-1: none),
//                out[2 k + 1] = its score (an inactive leaf can be returned: score = MIN).
//   state[i] = tree[i].p after the script, i < tree.size() (returned; state may be null).
namespace {
struct RefCoor {
  pair<int, int> x;
  int score, pos;
  bool operator<(const RefCoor &a) const { return x < a.x; }
};
}  // namespace
int ref_segtree_script(const int *pts, int n, const int *ops, int nops, int *out, int *state, int state_cap) {
  if (n < 2) return -1;
  vector<RefCoor> ys;
  for (int i = 0; i < n; i++) ys.push_back({{pts[2 * i], pts[2 * i + 1]}, SegmentTree<RefCoor>::MIN, i});
  SegmentTree<RefCoor> tree(ys);
  for (int k = 0; k < nops; k++) {
    const int *o = ops + 5 * k;
    out[2 * k] = out[2 * k + 1] = -2;
    if (o[0] == 0) {
      tree.activate({o[1], o[2]}, o[3]);
    } else if (o[0] == 1) {
      tree.deactivate({o[1], o[2]});
    } else {
      const int j = tree.rmq({o[1], o[2]}, {o[3], o[4]});
      out[2 * k] = j == -1 ? -1 : ys[j].pos;
      out[2 * k + 1] = j == -1 ? 0 : ys[j].score;
    }
  }
  const int size = (int)tree.tree.size();
  if (state)
    for (int i = 0; i < size && i < state_cap; i++) state[i] = tree.tree[i].p;
  return size;
}

// fmt::format("{}", double) of the reference's vendored fmt (extern/format.cc, 4.0.1): how `stats generate` prints its
// floating-point columns (src/stats_main.cc:315-334)
int ref_fmt_double(double x, char *out, size_t cap) { return emit(fmt::format("{}", x), out, cap); }

// CLI scoring overrides (src/align_main.cc:343-352 assign these statics; src/align.cc:84-86,343-456 read them)
int ref_set_scoring(int match, int mismatch, int gap_open, int gap_extend) {
  Globals::Align::MATCH = match;
  Globals::Align::MISMATCH = mismatch;
  Globals::Align::GAP_OPEN = gap_open;
  Globals::Align::GAP_EXTEND = gap_extend;
  return 0;
}
}
