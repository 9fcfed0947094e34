// Seed anchors and sparse chaining on the host (what generate_anchors and chain_anchors of the reference compute,
// src/chain.cc:24-199, with the range-maximum structure of src/segment.{h,tpp}).
//
// The stage driver takes its anchors from the GPU (anchors.hip); this file is the host path for what the device
// kernels do not cover (k > 11, sequences of 4 Mb and more) and for chaining.  Both are checked against the
// brute-force definitions in tests/bruteforce.py; the reference's own code cannot be compiled here (Boost).
//
//   anchors   sort-merge join of the two k-mer lists instead of a hash map of position lists: the reference
//             k-mers are packed with their positions into 64-bit keys and sorted once; every query k-mer finds its
//             positions with one binary search, in ascending order, which is the order the reference visits them in.
//   chaining  a sweep over the anchors' start and end points in query order with a range-maximum structure over the
//             reference end points.  WHICH of several equally good predecessors an anchor gets is decided by that
//             structure's shape and update rules (src/segment.tpp:62,89,128), so the structure here is the same
//             priority search tree -- flat arrays, no recursion -- and its choices are the reference's.
#include <algorithm>
#include <climits>
#include <functional>
#include <cstdlib>

#include "sedef_host.h"

namespace sdfh {

// ---- anchors ---------------------------------------------------------------------------------------------
namespace {
// the sequence in upper case and the running count of upper-case input letters
struct ScannedSeq {
  std::string up;
  std::vector<int> upper_before;  // upper_before[i] = upper-case letters among the first i bases
  explicit ScannedSeq(const std::string &s) : up(s.size(), ' '), upper_before(s.size() + 1, 0) {
    for (size_t i = 0; i < s.size(); i++) {
      const unsigned char c = (unsigned char)s[i];
      up[i] = (char)toupper(c);
      upper_before[i + 1] = upper_before[i] + (c >= 'A' && c <= 'Z');
    }
  }
};

// calls f(position, kmer) for every k-mer window without an N, in ascending position: the rolling 2-bit hash and the
// position of the last N (src/chain.cc:29-39, :49-58)
template <typename F>
void for_each_kmer(const std::string &s, int k, F f) {
  const uint32_t mask = (uint32_t)((1ull << (2 * k)) - 1);
  uint32_t h = 0;
  int last_n = -k;
  for (int i = 0; i < (int)s.size(); i++) {
    if (toupper((unsigned char)s[i]) == 'N') last_n = i;
    h = ((h << 2) | (uint32_t)hash_dna(s[i])) & mask;
    if (i >= k - 1 && last_n < i - k + 1) f(i - k + 1, h);
  }
}
}  // namespace

std::vector<Anchor> generate_anchors(const std::string &query, const std::string &ref, const Hit &orig, int kmer_size) {
  std::vector<uint64_t> rk;  // (k-mer << 32) | position
  rk.reserve(ref.size());
  for_each_kmer(ref, kmer_size, [&](int pos, uint32_t h) { rk.push_back(((uint64_t)h << 32) | (uint32_t)pos); });
  std::sort(rk.begin(), rk.end());

  const ScannedSeq Q(query), R(ref);
  const int qn = (int)query.size(), rn = (int)ref.size();
  // covered[d]: first query position of diagonal d (= qn + r - q) not yet inside an anchor (src/chain.cc:42,70-72)
  std::vector<int> covered((size_t)qn + rn, -1);
  const bool same_chr = orig.query->name == orig.ref->name && orig.query->is_rc == orig.ref->is_rc;
  const int self_shift = orig.ref_start - orig.query_start;  // diagonal of "the same genome position" (:67-69)
  std::vector<Anchor> out;
  for_each_kmer(query, kmer_size, [&](int q, uint32_t h) {
    auto lo = std::lower_bound(rk.begin(), rk.end(), (uint64_t)h << 32);
    auto hi = lo;
    while (hi != rk.end() && (*hi >> 32) == h) ++hi;
    if (lo == hi || hi - lo >= 1000) return;  // frequent k-mers are not seeds (:61)
    for (auto it = lo; it != hi; ++it) {
      const int r = (int)(uint32_t)*it;
      if (same_chr && abs(self_shift + r - q) <= kmer_size) continue;
      int &cov = covered[(size_t)(qn + r - q)];
      if (q < cov) continue;
      int len = 0;  // maximal exact match, case-insensitive, stopping at an N (:76-85)
      while (q + len < qn && r + len < rn && Q.up[q + len] == R.up[r + len] && Q.up[q + len] != 'N') ++len;
      if (len < kmer_size) continue;
      const bool any_upper = Q.upper_before[q + len] - Q.upper_before[q] + R.upper_before[r + len] - R.upper_before[r] > 0;
      out.push_back(Anchor{q, r, len, any_upper ? 1 : 0});  // (the reference adds flags into a bool: 0 or 1, :74,84)
      cov = q + len;
    }
  });
  return out;
}

// ---- chaining --------------------------------------------------------------------------------------------
namespace {
const int kInactive = INT_MIN;

inline int64_t key_of(int coord, int idx) { return (int64_t)coord * ((int64_t)1 << 32) + idx; }

// Sorts keys of the form key_of(coord, idx) whose idx ascend in the input (so a STABLE sort by coord alone is the sort by
// the whole key): counting passes over 11-bit digits of the coordinate -- two for sequences below 4 Mb -- instead of a
// comparison sort; the three sorts of a pair's anchors were a third of chain_anchors.  Short inputs and negative
// coordinates keep std::sort.
void sort_by_coord(std::vector<int64_t> &v, std::vector<int64_t> &tmp, int64_t max_coord) {
  const size_t n = v.size();
  if (n < 96 || max_coord < 0 || max_coord >= ((int64_t)1 << 31)) {
    std::sort(v.begin(), v.end());
    return;
  }
  tmp.resize(n);
  int64_t *src = v.data(), *dst = tmp.data();
  for (int shift = 32; (max_coord >> (shift - 32)) != 0; shift += 11) {
    uint32_t count[2049] = {0};
    for (size_t k = 0; k < n; k++) count[((uint64_t)src[k] >> shift & 2047u) + 1]++;
    for (int d = 0; d < 2048; d++) count[d + 1] += count[d];
    for (size_t k = 0; k < n; k++) dst[count[(uint64_t)src[k] >> shift & 2047u]++] = src[k];
    std::swap(src, dst);
  }
  if (src != v.data()) v.swap(tmp);
}

// Priority search tree over points sorted by key: node i covers a range of points, leaves hold one point, and `top` of a
// node names the leaf (by node index) of the best active point below it that no ancestor has claimed.  Heap layout, split
// rule and the three tie rules are those of src/segment.tpp.
//
// Round 6: the sweep below spends most of a pair's host time in this structure (2.1 million anchors in the chr1-sized stage
// run), so a node is ONE 16-byte record -- its `top`, and either the largest key of its left child's range (what every descent
// compares with) or, for a leaf, its own key and score -- instead of five parallel arrays, a point finds its leaf through a
// table instead of a descent from the root, and points are named by their rank.  The operations and their order are
// unchanged (tests/test_chain_oracle.py replays scripts on this class and on the reference's SegmentTree).
class RangeMax {
 public:
  std::vector<int64_t> key;  // per point, ascending
  explicit RangeMax(std::vector<int64_t> sorted_keys) : key(std::move(sorted_keys)), leaf_(key.size(), -1) {
    const int n = (int)key.size();
    int bits = 0;
    for (unsigned v = (unsigned)n - 1u; n > 0 && v; v >>= 1) bits++;
    const int size = 2 << (n > 0 ? bits : 0);  // twice the next power of two (src/segment.tpp:18-19)
    node_.assign((size_t)size + 1, Node{0, -1, kInternal});  // (+1: the right child of the last node is tested, never entered)
    size_ = size;
    // node i covers [lo, hi); children split at (lo + hi + 1) / 2 (src/segment.tpp:184)
    std::vector<int> lo((size_t)size, 0), hi((size_t)size, 0);
    std::vector<int64_t> reach((size_t)size, 0);
    if (n > 0) hi[0] = n;
    for (int i = 0; i < size; i++) {
      if (hi[i] <= lo[i]) continue;
      reach[i] = key[(size_t)hi[i] - 1];
      if (hi[i] - lo[i] == 1) {
        node_[i].k = key[(size_t)lo[i]];
        node_[i].score = kInactive;
        leaf_[(size_t)lo[i]] = i;
        continue;
      }
      const int mid = (lo[i] + hi[i] + 1) / 2;
      lo[2 * i + 1] = lo[i];
      hi[2 * i + 1] = mid;
      lo[2 * i + 2] = mid;
      hi[2 * i + 2] = hi[i];
    }
    for (int i = 0; 2 * i + 1 < size; i++)
      if (node_[i].score == kInternal) node_[i].k = reach[2 * i + 1];
    point_of_leaf_.assign((size_t)size, -1);
    for (int p = 0; p < n; p++) point_of_leaf_[(size_t)leaf_[(size_t)p]] = p;
  }

  int score_of(int pt) const { return node_[leaf_[(size_t)pt]].score; }
  int point_of_key(int64_t k) const { return (int)(std::lower_bound(key.begin(), key.end(), k) - key.begin()); }

  // point with the largest score among the active points with lo <= key <= hi, or -1.  Among equal scores: a
  // subtree's own top before anything below it, the left subtree before the right one (src/segment.tpp:29-66).
  int best_in(int64_t lo, int64_t hi) const {
    int best_leaf = -1, best_score = 0;
    int stack[96], sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const int i = stack[--sp];
      const Node &nd = node_[i];
      int cand = -1;
      if (nd.score != kInternal) {
        if (lo <= nd.k && nd.k <= hi) cand = i;
      } else if (nd.top != -1) {
        const int64_t tk = node_[nd.top].k;
        if (lo <= tk && tk <= hi) {
          cand = nd.top;
        } else if (hi <= nd.k) {
          stack[sp++] = 2 * i + 1;
        } else if (lo > nd.k) {
          stack[sp++] = 2 * i + 2;
        } else {
          stack[sp++] = 2 * i + 2;  // (popped after the left one: candidates arrive left to right)
          stack[sp++] = 2 * i + 1;
        }
      }
      if (cand != -1 && (best_leaf == -1 || node_[cand].score > best_score)) {
        best_leaf = cand;
        best_score = node_[cand].score;
      }
    }
    return best_leaf == -1 ? -1 : point_of_leaf_[(size_t)best_leaf];
  }

  void activate(int pt, int s) {  // src/segment.tpp:76-103: a newcomer takes the place of an equal score
    int leaf = leaf_[(size_t)pt];
    node_[leaf].score = s;
    for (int i = 0; i < size_ && leaf != -1;) {
      Node &nd = node_[i];
      if (nd.top == -1 || node_[leaf].score >= node_[nd.top].score) std::swap(nd.top, leaf);
      if (leaf == -1) break;
      i = 2 * i + 1 + (node_[leaf].k > split_of(nd));
    }
  }

  void deactivate(int pt) {  // src/segment.tpp:105-146: the hole is filled from below, left child on equal scores
    int leaf = leaf_[(size_t)pt];
    const int64_t k = node_[leaf].k;
    node_[leaf].score = kInactive;
    for (int i = 0; i < size_ && node_[i].top != -1;) {
      Node &nd = node_[i];
      if (nd.top != leaf) {
        i = 2 * i + 1 + (k > split_of(nd));
      } else if (nd.score != kInternal) {
        nd.top = -1;
      } else {
        const int l = 2 * i + 1, r = 2 * i + 2;
        const int tl = node_[l].top, tr = r < size_ ? node_[r].top : -1;
        const bool right = tr != -1 && (tl == -1 || node_[tr].score > node_[tl].score);
        nd.top = leaf = right ? tr : tl;
        i = right ? r : l;
      }
    }
  }

  int dump_tops(int *state, int cap) const {  // (test hook: the node each node's best point sits in)
    for (int i = 0; i < size_ && i < cap; i++) state[i] = node_[(size_t)i].top;
    return size_;
  }

 private:
  static const int kInternal = INT_MIN + 1;  // `score` of a node that is not a leaf (no chain scores that low)
  struct Node {          // 16 bytes: four nodes a cache line
    int64_t k;           // a leaf: its point's key; an inner node: the largest key below its left child (0: none)
    int32_t top;         // leaf node of the best unclaimed active point below, or -1
    int32_t score;       // leaf: its point's score (kInactive while not active); kInternal otherwise
  };
  // (what the descent of activate compares with: an inner node's split; a leaf has no children -- 0, as the reference's
  // `reach` of a childless node)
  static int64_t split_of(const Node &nd) { return nd.score != kInternal ? 0 : nd.k; }
  std::vector<Node> node_;
  std::vector<int> leaf_, point_of_leaf_;
  int size_ = 0;
};
}  // namespace

// Test hook: replays a script of activate / deactivate / best_in calls on a RangeMax over the given points -- the same
// script format as oracle/ref_align_driver.cc: ref_segtree_script, which drives the reference's own SegmentTree class;
// the tests compare every answer and the tree's final state.  pts[2 i + 1] must be i.
int rangemax_script(const int *pts, int n, const int *ops, int nops, int *out, int *state, int state_cap) {
  std::vector<int64_t> keys((size_t)n);
  for (int i = 0; i < n; i++) keys[(size_t)i] = key_of(pts[2 * i], pts[2 * i + 1]);
  std::sort(keys.begin(), keys.end());
  RangeMax tree(keys);
  for (int k = 0; k < nops; k++) {
    const int *o = ops + 5 * k;
    out[2 * k] = out[2 * k + 1] = -2;
    if (o[0] == 0) {
      tree.activate(tree.point_of_key(key_of(o[1], o[2])), o[3]);
    } else if (o[0] == 1) {
      tree.deactivate(tree.point_of_key(key_of(o[1], o[2])));
    } else {
      const int pt = tree.best_in(key_of(o[1], o[2]), key_of(o[3], o[4]));
      out[2 * k] = pt == -1 ? -1 : (int)(uint32_t)tree.key[(size_t)pt];
      out[2 * k + 1] = pt == -1 ? 0 : tree.score_of(pt);
    }
  }
  return tree.dump_tops(state, state_cap);
}

std::pair<std::vector<int>, std::vector<std::pair<int, bool>>> chain_anchors(std::vector<Anchor> &anchors,
                                                                            const Params &P) {
  const int n = (int)anchors.size();
  std::vector<int> path;
  std::vector<std::pair<int, bool>> boundaries{{0, false}};
  if (n == 0) return {path, boundaries};

  // sweep events in (query coordinate, anchor) order: an anchor's start looks for a predecessor, its end makes it
  // available as one (src/chain.cc:112-135).  The anchors arrive in the order generate_anchors emits them -- ascending query
  // position --, so the start events (q, i) are already in order: only the end events are sorted, and the sweep merges the two
  // lists as it goes (an unsorted input sorts its starts too).
  std::vector<int64_t> starts((size_t)n), ends((size_t)n), sorted_r((size_t)n);
  int far_q = 0, far_r = 0;
  bool starts_sorted = true;
  for (int i = 0; i < n; i++) {
    const Anchor &a = anchors[i];
    starts[(size_t)i] = key_of(a.q, i);
    ends[(size_t)i] = key_of(a.q + a.l, i);
    sorted_r[(size_t)i] = key_of(a.r + a.l - 1, i);
    starts_sorted = starts_sorted && (i == 0 || anchors[i - 1].q <= a.q);
    far_q = std::max(far_q, a.q + a.l);
    far_r = std::max(far_r, a.r + a.l);
  }
  std::vector<int64_t> scratch;
  if (!starts_sorted) sort_by_coord(starts, scratch, far_q);
  sort_by_coord(ends, scratch, far_q);
  sort_by_coord(sorted_r, scratch, far_r);
  std::vector<int> point_of((size_t)n);  // anchor -> its point of the tree (the rank of its reference end)
  for (int p = 0; p < n; p++) point_of[(size_t)(uint32_t)sorted_r[(size_t)p]] = p;
  RangeMax tree(std::move(sorted_r));
  // (point of the tree -> anchor: the low half of its key)

  std::vector<int> best(n, 0), pred(n, -1);
  size_t expire = 0;  // end events before this one have been checked for expiry
  for (size_t si = 0, ei = 0; si < (size_t)n || ei < (size_t)n;) {
    if (si == (size_t)n || (ei < (size_t)n && ends[ei] < starts[si])) {
      // end point; stored score = chain score minus the way to the far corner (:175-176)
      const int i = (int)(uint32_t)ends[ei++];
      const Anchor &a = anchors[i];
      tree.activate(point_of[(size_t)i], best[i] - ((far_q + 1 - (a.q + a.l)) + (far_r + 1 - (a.r + a.l))));
      continue;
    }
    const int i = (int)(uint32_t)starts[si++];
    const Anchor &a = anchors[i];
    // anchors that ended more than max_chain_gap before this start are no predecessors any more (:142-152); such an end
    // lies before this start in the event order, so it has been activated
    for (; expire < ei; expire++) {
      const int t = (int)(uint32_t)ends[expire];
      if (a.q - (int)(ends[expire] >> 32) <= P.max_chain_gap) break;
      tree.deactivate(point_of[(size_t)t]);
    }
    const int w = P.match_chain_score * a.has_u + (P.match_chain_score / 2) * (a.l - a.has_u);
    best[i] = w;
    const int pt = tree.best_in(key_of(a.r - P.max_chain_gap, 0), key_of(a.r - 1, n));
    if (pt != -1 && tree.score_of(pt) != kInactive) {
      const int j = (int)(uint32_t)tree.key[(size_t)pt];
      const Anchor &p = anchors[j];
      const int with = w + best[j] - ((a.q - (p.q + p.l)) + (a.r - (p.r + p.l)));
      if (with > 0) {
        best[i] = with;
        pred[i] = j;
      }
    }
  }
  // chains, best first (ties: the later anchor first), each from its last anchor backwards until it meets an anchor
  // an earlier chain took (src/chain.cc:179-197)
  std::vector<int64_t> order((size_t)n);  // (score, anchor) in one word: sorted as numbers
  int best_max = 0;
  for (int i = 0; i < n; i++) {
    order[(size_t)i] = key_of(best[i], i);
    best_max = std::max(best_max, best[i]);
  }
  sort_by_coord(order, scratch, best_max);  // (ascending, ties by ascending anchor: read backwards)
  std::reverse(order.begin(), order.end());
  path.reserve((size_t)n);
  std::vector<char> taken(n, 0);
  for (const int64_t ord : order) {
    const int head = (int)(uint32_t)ord;
    if (taken[head]) continue;
    int upper = 0;
    for (int i = head; i != -1 && !taken[i]; i = pred[i]) {
      path.push_back(i);
      upper += anchors[i].has_u;
      taken[i] = 1;
    }
    boundaries.push_back({(int)path.size(), upper != 0});
  }
  return {path, boundaries};
}

}  // namespace sdfh
