// TEST INFRASTRUCTURE: the REFERENCE's own Hit / Alignment objects behind integer handles, so that a model of
// fast_align / refine_chains / the stage driver written elsewhere (tests/bruteforce.py: StageModel -- control flow
// only) can have every alignment built, merged, extended and printed by the reference's classes themselves
// (src/align.cc, src/hit.cc; compiled unmodified into _ref/libref_align.so by the Makefile's `refalign`).
// src/chain.cc, src/refine.cc and src/align_main.cc cannot be compiled in this image (Boost.ICL through
// src/search.h:22-23); what they do BETWEEN calls into these classes is what the model restates.
//
// Nothing here computes: each function is one constructor / method call of the reference, at the place the cited
// reference line makes it.
#include <algorithm>
#include <cassert>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "align.h"
#include "fasta.h"
#include "hash.h"
#include "hit.h"
#include "segment.h"

using namespace std;

namespace {
struct Table {
  string query, ref;
  shared_ptr<Sequence> query_ptr, ref_ptr;
  vector<Hit> hits;
};
Table g_tab;

// a point type shaped like chain_anchors' own (`Coor`, src/chain.cc:106-110: local to a function of a file that cannot
// be compiled here), for the reference's SegmentTree<T> (src/segment.h, src/segment.tpp)
struct TreeCoor {
  pair<int, int> x;
  int score, pos;
  bool operator<(const TreeCoor &a) const { return x < a.x; }
};
vector<TreeCoor> g_ys;
unique_ptr<SegmentTree<TreeCoor>> g_tree;

int emit(const string &s, char *buf, size_t cap) {
  if (s.size() + 1 > cap) return -2;
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}
bool bad(int id) { return id < 0 || id >= (int)g_tab.hits.size(); }
}  // namespace

extern "C" {

// fast_align's two shared sequences (src/chain.cc:207-208); drops every handle of the previous pair
int ref_tab_open(const char *query, const char *ref) {
  g_tab.hits.clear();
  g_tab.query = query;
  g_tab.ref = ref;
  g_tab.query_ptr = make_shared<Sequence>("QRY", g_tab.query);
  g_tab.ref_ptr = make_shared<Sequence>("REF", g_tab.ref);
  return 0;
}

// One first-round chain (src/chain.cc:243-258): Hit a{query_ptr, qlo, qhi, ref_ptr, rlo, rhi, up}; a.aln =
// Alignment(query, ref, anchors, guide); update_from_alignment(a).  anchors: n x {q, r, l, has_u}; guide: indices.
int ref_tab_chain_hit(const int *anchors, int n, const int *guide, int ng, int qlo, int qhi, int rlo, int rhi, int up) {
  vector<Anchor> an(n);
  for (int i = 0; i < n; i++) an[i] = Anchor{anchors[4 * i], anchors[4 * i + 1], anchors[4 * i + 2], anchors[4 * i + 3]};
  vector<int> gi(guide, guide + ng);
  Hit a{g_tab.query_ptr, qlo, qhi, g_tab.ref_ptr, rlo, rhi, up};
  a.aln = Alignment(g_tab.query, g_tab.ref, an, gi);
  update_from_alignment(a);
  g_tab.hits.push_back(a);
  return (int)g_tab.hits.size() - 1;
}

// out: query_start, query_end, ref_start, ref_end, jaccard, matches, mismatches, gap_bases, gaps, span
int ref_tab_get(int id, int *out) {
  if (bad(id)) return -1;
  const Hit &h = g_tab.hits[id];
  out[0] = h.query_start;
  out[1] = h.query_end;
  out[2] = h.ref_start;
  out[3] = h.ref_end;
  out[4] = h.jaccard;
  out[5] = h.aln.matches();
  out[6] = h.aln.mismatches();
  out[7] = h.aln.gap_bases();
  out[8] = h.aln.gaps();
  out[9] = h.aln.span();
  return 0;
}

int ref_tab_cigar(int id, char *buf, size_t cap) {
  if (bad(id)) return -1;
  return emit(g_tab.hits[id].aln.cigar_string(), buf, cap);
}

// src/refine.cc:172-173: prev->aln.merge(cur.aln, qseq, rseq); update_from_alignment(*prev)
int ref_tab_merge(int prev, int cur) {
  if (bad(prev) || bad(cur)) return -1;
  g_tab.hits[prev].aln.merge(g_tab.hits[cur].aln, g_tab.query, g_tab.ref);
  update_from_alignment(g_tab.hits[prev]);
  return 0;
}

// src/refine.cc:164-165,181-183: Hit{anchors.front().query, qlo, qhi, anchors.front().ref, rlo, rhi}; hit.aln =
// Alignment(hit.query->seq, hit.ref->seq, guide, side); update_from_alignment(hit).  guide: handles, copied in order
// like guide.push_back(*prev) (src/refine.cc:175,179).
int ref_tab_guide_hit(const int *guide, int ng, int qlo, int qhi, int rlo, int rhi, int side) {
  vector<Hit> g;
  for (int i = 0; i < ng; i++) {
    if (bad(guide[i])) return -1;
    g.push_back(g_tab.hits[guide[i]]);
  }
  auto hit = Hit{g_tab.query_ptr, qlo, qhi, g_tab.ref_ptr, rlo, rhi};
  hit.aln = Alignment(hit.query->seq, hit.ref->seq, g, side);
  update_from_alignment(hit);
  g_tab.hits.push_back(hit);
  return (int)g_tab.hits.size() - 1;
}

// The stage driver's in-place remap of one result (src/align_main.cc:314-327): the caller computes the four
// coordinates; names and the rc flag go into the Sequence objects every hit of the pair shares.
int ref_tab_remap(int id, int qs, int qe, int rs, int re, const char *qname, const char *rname, int ref_is_rc) {
  if (bad(id)) return -1;
  Hit &hh = g_tab.hits[id];
  hh.query_start = qs;
  hh.query_end = qe;
  hh.ref_start = rs;
  hh.ref_end = re;
  if (ref_is_rc) hh.ref->is_rc = true;
  hh.query->name = qname;
  hh.ref->name = rname;
  return 0;
}

// hh.to_bed(false) (src/align_main.cc:330, src/hit.cc:134-196)
int ref_tab_to_bed(int id, char *buf, size_t cap) {
  if (bad(id)) return -1;
  return emit(g_tab.hits[id].to_bed(false), buf, cap);
}

// h.to_bed(0) of a seed hit (src/align_main.cc:330).  Hit::from_bed (src/hit.cc:29-63) needs split() of the
// Boost-dependent src/util.cc, so the fields it fills are given one by one: the caller has split the line.
int ref_seed_to_bed(const char *qname, int q_rc, int qs, int qe, const char *rname, int r_rc, int rs, int re, const char *name,
                    const char *comment, int jaccard, char *buf, size_t cap) {
  Hit h{make_shared<Sequence>(qname, "", false), qs, qe, make_shared<Sequence>(rname, "", false), rs, re, jaccard, name,
        comment, {}};
  h.query->is_rc = q_rc != 0;  // (the constructor's rc() of an empty string is the empty string)
  h.ref->is_rc = r_rc != 0;
  if (h.query->is_rc) return -3;  // to_bed asserts !query->is_rc (src/hit.cc:136)
  return emit(h.to_bed(0), buf, cap);
}

// ---- the reference's SegmentTree, live: chain_anchors' sweep (src/chain.cc:138-178) decides each call from the answer
// of the previous ones, so the model holds one tree and calls it as it goes.
// src/chain.cc:113-132: ys[i] = {{r + l - 1, i}, MIN, i}; SegmentTree<Coor> tree(ys)  (n >= 2: the constructor takes clz(n - 1))
int ref_tree_new(const int *pts, int n) {
  if (n < 2) return -1;
  g_tree.reset();
  g_ys.clear();
  for (int i = 0; i < n; i++) g_ys.push_back({{pts[2 * i], pts[2 * i + 1]}, SegmentTree<TreeCoor>::MIN, i});
  g_tree.reset(new SegmentTree<TreeCoor>(g_ys));
  return 0;
}
int ref_tree_activate(int x0, int x1, int score) {
  g_tree->activate({x0, x1}, score);
  return 0;
}
int ref_tree_deactivate(int x0, int x1) {
  g_tree->deactivate({x0, x1});
  return 0;
}
// j = tree.rmq(p, q); returns -1, or -2 when ys[j].score == MIN (src/chain.cc:160), else ys[j].pos
int ref_tree_rmq(int p0, int p1, int q0, int q1) {
  const int j = g_tree->rmq({p0, p1}, {q0, q1});
  if (j == -1) return -1;
  if (g_ys[j].score == SegmentTree<TreeCoor>::MIN) return -2;
  return g_ys[j].pos;
}
}
