// Seed anchors and sparse chaining (restates reference src/chain.cc:24-199 and the priority-search
// "segment tree" of src/segment.{h,tpp}, whose tie-breaking decides which predecessor a chain takes).
#include <algorithm>
#include <cassert>
#include <cctype>
#include <climits>
#include <cstdlib>
#include <list>
#include <unordered_map>

#include "sedef_host.h"

namespace sdfh {

std::vector<Anchor> generate_anchors(const std::string &query, const std::string &ref, const Hit &orig,
                                     int kmer_size) {  // src/chain.cc:24-101
  const uint32_t MASK = (1u << (2 * kmer_size)) - 1;
  std::unordered_map<uint32_t, std::vector<int>> ref_hashes;  // positions in ascending order
  int last_n = -kmer_size;
  uint32_t h = 0;
  for (int i = 0; i < (int)ref.size(); i++) {
    if (toupper(ref[i]) == 'N') last_n = i;
    h = ((h << 2) | (uint32_t)hash_dna(ref[i])) & MASK;
    if (i < kmer_size - 1) continue;
    if (last_n >= (i - kmer_size + 1)) continue;
    ref_hashes[h].push_back(i - kmer_size + 1);
  }

  std::vector<int> slide(query.size() + ref.size(), -1);
  std::vector<Anchor> anchors;
  const bool same_chr = orig.query->name == orig.ref->name && orig.query->is_rc == orig.ref->is_rc;

  last_n = -kmer_size, h = 0;
  for (int i = 0; i < (int)query.size(); i++) {
    if (toupper(query[i]) == 'N') last_n = i;
    h = ((h << 2) | (uint32_t)hash_dna(query[i])) & MASK;
    if (i < kmer_size - 1) continue;
    if (last_n >= (i - kmer_size + 1)) continue;
    auto it = ref_hashes.find(h);
    if (it == ref_hashes.end() || it->second.size() >= 1000) continue;
    const int q = i - kmer_size + 1;
    const int off = (int)query.size();
    for (int r : it->second) {
      if (same_chr && abs(orig.ref_start + r - (orig.query_start + q)) <= kmer_size) continue;
      const int d = off + r - q;
      if (q >= slide[d]) {
        bool has_u = false;  // "any uppercase base": the reference accumulates into a bool
        int len;
        for (len = 0; q + len < (int)query.size() && r + len < (int)ref.size(); len++) {
          if (toupper(query[q + len]) == 'N' || toupper(ref[r + len]) == 'N') break;
          if (toupper(query[q + len]) != toupper(ref[r + len])) break;
          has_u = has_u || isupper((unsigned char)query[q + len]) || isupper((unsigned char)ref[r + len]);
        }
        if (len >= kmer_size) {
          anchors.push_back(Anchor{q, r, len, has_u ? 1 : 0});
          slide[d] = q + len;
        }
      }
    }
  }
  return anchors;
}

namespace {
const int TREE_MIN = INT_MIN;

struct Coor {  // src/chain.cc:106-110
  std::pair<int, int> x;
  int score, pos;
  bool operator<(const Coor &o) const { return x < o.x; }
};

// Priority search tree over the anchors' reference end points (src/segment.tpp).  The array layout
// (heap indices), the split keys `h`, the "best active point" slot `p` of every node and the order of
// the comparisons are what make rmq's answer unique among equal scores, so they are kept as they are.
struct PrioTree {
  struct Node {
    int p = -1, a = -1;
    std::pair<int, int> h;
  };
  std::vector<Node> tree;
  std::vector<Coor> &pts;

  explicit PrioTree(std::vector<Coor> &a) : pts(a) {
    std::sort(pts.begin(), pts.end());
    const unsigned n1 = (unsigned)pts.size() - 1u;
    // 1 << (32 - clz(n-1)); clz(0) is taken as 32 (lzcnt), i.e. one point -> size 1
    int bits = 0;
    for (unsigned v = n1; v; v >>= 1) bits++;
    const int size = pts.empty() ? 1 : (1 << bits);
    tree.resize((size_t)size << 1);
    int tree_i = 0;
    build(0, 0, (int)pts.size(), tree_i);
  }

  int build(int i, int s, int e, int &tree_i) {  // src/segment.tpp:172-192
    if (i >= (int)tree.size()) return -1;
    if (s + 1 == e) {
      tree[i].p = -1;
      tree[i].a = tree_i;
      tree[i].h = pts[tree_i].x;
      pts[tree_i].score = TREE_MIN;
      tree_i++;
      return i;
    }
    const int bnd = (s + e + 1) / 2;
    const int a = build(2 * i + 1, s, bnd, tree_i);
    const int b = build(2 * i + 2, bnd, e, tree_i);
    tree[i].p = -1;
    tree[i].a = -1;
    tree[i].h = tree[2 * i + 1 + (2 * i + 2 < (int)tree.size())].h;
    return std::max(a, std::max(i, b));
  }

  int rmq(const std::pair<int, int> &p, const std::pair<int, int> &q, int i) const {  // :29-66
    if (i >= (int)tree.size()) return -1;
    if (tree[i].a != -1) {
      return (p <= pts[tree[i].a].x && pts[tree[i].a].x <= q) ? i : -1;
    }
    const int pv = tree[i].p;
    if (pv == -1) return -1;
    if (p <= pts[tree[pv].a].x && pts[tree[pv].a].x <= q) return pv;
    if (q <= tree[2 * i + 1].h) return rmq(p, q, 2 * i + 1);
    if (p > tree[2 * i + 1].h) return rmq(p, q, 2 * i + 2);
    const int m1 = rmq(p, q, 2 * i + 1);
    const int m2 = rmq(p, q, 2 * i + 2);
    if (m1 == -1) return m2;
    if (m2 == -1) return m1;
    return pts[tree[m1].a].score >= pts[tree[m2].a].score ? m1 : m2;
  }
  int rmq(const std::pair<int, int> &p, const std::pair<int, int> &q) const {
    const int i = rmq(p, q, 0);
    return i == -1 ? -1 : tree[i].a;
  }

  int find_leaf(const std::pair<int, int> &q) const {
    int leaf = 0;
    while (leaf < (int)tree.size() && (tree[leaf].a == -1 || q != pts[tree[leaf].a].x))
      leaf = 2 * leaf + 1 + (q > tree[2 * leaf + 1].h);
    return leaf;
  }

  void activate(const std::pair<int, int> &q, int score) {  // :76-103
    int leaf = find_leaf(q);
    pts[tree[leaf].a].score = score;
    for (int i = 0; i < (int)tree.size();) {
      if (tree[i].p == -1 || pts[tree[leaf].a].score >= pts[tree[tree[i].p].a].score) std::swap(tree[i].p, leaf);
      if (leaf == -1) break;
      i = 2 * i + 1 + (pts[tree[leaf].a].x > tree[2 * i + 1].h);
    }
  }

  void deactivate(const std::pair<int, int> &q) {  // :105-146
    int leaf = find_leaf(q);
    pts[tree[leaf].a].score = TREE_MIN;
    for (int i = 0; i < (int)tree.size();) {
      if (tree[i].p == -1) break;
      if (tree[i].p == leaf) {
        if (tree[i].a != -1) {
          tree[i].p = -1;
        } else if (2 * i + 2 < (int)tree.size() && tree[2 * i + 2].p != -1 &&
                   (tree[2 * i + 1].p == -1 ||
                    pts[tree[tree[2 * i + 2].p].a].score > pts[tree[tree[2 * i + 1].p].a].score)) {
          tree[i].p = leaf = tree[2 * i + 2].p;
          i = 2 * i + 2;
        } else {
          tree[i].p = leaf = tree[2 * i + 1].p;
          i = 2 * i + 1;
        }
      } else {
        i = 2 * i + 1 + (q > tree[2 * i + 1].h);
      }
    }
  }
};
}  // namespace

std::pair<std::vector<int>, std::vector<std::pair<int, bool>>> chain_anchors(std::vector<Anchor> &anchors,
                                                                            const Params &P) {
  // src/chain.cc:103-199
  std::vector<Coor> xs, ys;
  xs.reserve(2 * anchors.size());
  ys.reserve(anchors.size());
  int max_q = 0, max_r = 0;
  for (int i = 0; i < (int)anchors.size(); i++) {
    const Anchor &a = anchors[i];
    xs.push_back({{a.q, i}, TREE_MIN, i});
    xs.push_back({{a.q + a.l, i}, TREE_MIN, i});
    ys.push_back({{a.r + a.l - 1, i}, TREE_MIN, i});
    max_q = std::max(max_q, a.q + a.l);
    max_r = std::max(max_r, a.r + a.l);
  }
  std::vector<int> path;
  std::vector<std::pair<int, bool>> boundaries{{0, 0}};
  if (anchors.empty()) return {path, boundaries};

  std::sort(xs.begin(), xs.end());
  PrioTree tree(ys);

  std::vector<int> prev(anchors.size(), -1);
  std::vector<std::pair<int, int>> dp(anchors.size());
  for (int i = 0; i < (int)dp.size(); i++) dp[i] = {0, i};
  int deactivate_bound = 0;
  for (int xi = 0; xi < (int)xs.size(); xi++) {
    const int i = xs[xi].x.second;
    const Anchor &a = anchors[i];
    if (xs[xi].x.first == a.q) {  // start point
      while (deactivate_bound < xi) {
        const int t = xs[deactivate_bound].x.second;
        if (xs[deactivate_bound].x.first == anchors[t].q + anchors[t].l) {  // an end point
          if (a.q - (anchors[t].q + anchors[t].l) <= P.max_chain_gap) break;
          tree.deactivate({anchors[t].r + anchors[t].l - 1, t});
        }
        deactivate_bound++;
      }
      const int w = P.match_chain_score * a.has_u + (P.match_chain_score / 2) * (a.l - a.has_u);
      int j = tree.rmq({a.r - P.max_chain_gap, 0}, {a.r - 1, (int)anchors.size()});
      if (j != -1 && ys[j].score != TREE_MIN) {
        j = ys[j].pos;
        const Anchor &p = anchors[j];
        const int gap = (a.q - (p.q + p.l) + a.r - (p.r + p.l));
        if (w + dp[j].first - gap > 0) {
          dp[i].first = w + dp[j].first - gap;
          prev[i] = j;
        } else {
          dp[i].first = w;
        }
      } else {
        dp[i].first = w;
      }
    } else {  // end point: the anchor becomes available as a predecessor
      const int gap = (max_q + 1 - (a.q + a.l) + max_r + 1 - (a.r + a.l));
      tree.activate({a.r + a.l - 1, i}, dp[i].first - gap);
    }
  }
  std::sort(dp.begin(), dp.end(), std::greater<std::pair<int, int>>());

  path.reserve(anchors.size());
  std::vector<char> used(anchors.size(), 0);
  for (auto &m : dp) {
    int maxi = m.second;
    if (used[maxi]) continue;
    int has_u = 0;
    while (maxi != -1 && !used[maxi]) {
      path.push_back(maxi);
      has_u += anchors[maxi].has_u;
      used[maxi] = true;
      maxi = prev[maxi];
    }
    boundaries.push_back({(int)path.size(), has_u});  // int -> bool: "any uppercase anchor"
  }
  return {path, boundaries};
}

}  // namespace sdfh
