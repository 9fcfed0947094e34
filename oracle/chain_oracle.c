/* TEST INFRASTRUCTURE -- CPU oracle for anchor chaining (scope rows C2 / f3).  Not part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Restates, in plain C:
 *   - SegmentTree<T> of the reference (src/segment.h:21-56, src/segment.tpp:12-192): the priority search tree over the
 *     anchors' end coordinates in which chain_anchors looks for the best predecessor;
 *   - chain_anchors (src/chain.cc:103-199): the sweep over start / end events and the extraction of the chains.
 *
 * Pinning.  The tree is PINNED: tests/test_chain_oracle.py replays scripts of activate / deactivate / rmq calls -- random
 * ones full of equal scores and equal coordinates, and the exact call streams chain_anchors issues for tie-rich anchor
 * sets -- on the reference's own class (oracle/ref_align_driver.cc: ref_segtree_script, compiled from src/segment.h
 * where it lies) and on sdfo_segtree_script below: every returned point, every score and the final `p` pointer of
 * every node must be equal, live when the checkout is present and from tests/golden/segtree_kat.json.gz otherwise.
 * The sweep around it (sdfo_chain_anchors) is a restatement of a function that cannot be compiled here
 * (src/chain.cc:17 includes src/search.h:22-23 -> Boost.ICL): parity unpinned for those ninety lines; what it computes
 * is checked against the O(n^2) definition in tests/bruteforce.py (dp values, admissible links, chain order).
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int a, b; /* pair<int,int>, compared lexicographically */
} key2;
static int k_lt(key2 x, key2 y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }
static int k_le(key2 x, key2 y) { return !k_lt(y, x); }
static int k_eq(key2 x, key2 y) { return x.a == y.a && x.b == y.b; }

typedef struct {
  key2 x;
  int score, pos; /* `Coor` of src/chain.cc:106-110 */
} pt_t;
typedef struct {
  int p, a; /* p: node of the best active point below that no ancestor holds; a: point of a leaf, -1 otherwise */
  key2 h;   /* largest key of the left subtree (inclusive), or the leaf's key */
} node_t;
typedef struct {
  node_t *t;
  pt_t *pts;
  int size;
} tree_t;

#define SEG_MIN INT_MIN /* src/segment.h:23 */

static int pt_cmp(const void *x, const void *y) {
  const pt_t *p = (const pt_t *)x, *q = (const pt_t *)y;
  return k_lt(p->x, q->x) ? -1 : k_lt(q->x, p->x) ? 1 : 0;
}

/* src/segment.tpp:172-192: leaves are laid out left to right over the sorted points; an inner node's h is taken from
 * its RIGHT child when that child exists in the array, from the left one otherwise.  Returns the largest node touched. */
static int seg_init(tree_t *T, int i, int s, int e, int *next) {
  if (i >= T->size) return -1;
  if (s + 1 == e) {
    T->t[i].p = -1;
    T->t[i].a = *next;
    T->t[i].h = T->pts[*next].x;
    T->pts[*next].score = SEG_MIN;
    ++*next;
    return i;
  }
  const int bnd = (s + e + 1) / 2;
  const int l = seg_init(T, 2 * i + 1, s, bnd, next);
  const int r = seg_init(T, 2 * i + 2, bnd, e, next);
  T->t[i].p = -1;
  T->t[i].a = -1;
  T->t[i].h = T->t[2 * i + 1 + (2 * i + 2 < T->size)].h;
  int m = l > i ? l : i;
  return r > m ? r : m;
}

/* src/segment.tpp:12-27.  n == 1 makes the reference evaluate __builtin_clz(0) (undefined); any size works for one
 * point, two nodes are taken here. */
static int seg_build(tree_t *T, pt_t *pts, int n) {
  qsort(pts, (size_t)n, sizeof(pt_t), pt_cmp); /* (keys are unique: the order does not depend on the sort) */
  int size = 1;
  if (n > 1) size = 1 << (32 - __builtin_clz((unsigned)(n - 1)));
  T->size = size << 1;
  T->t = (node_t *)malloc((size_t)T->size * sizeof(node_t));
  if (!T->t) return -1;
  for (int i = 0; i < T->size; i++) { /* Point(): a = -1, p = -1, h = {} (src/segment.h:30) */
    T->t[i].p = T->t[i].a = -1;
    T->t[i].h.a = T->t[i].h.b = 0;
  }
  T->pts = pts;
  int next = 0;
  seg_init(T, 0, 0, n, &next);
  return 0;
}

/* src/segment.tpp:29-66: node of the best point with p <= x <= q below node i, or -1 */
static int seg_rmq(const tree_t *T, key2 p, key2 q, int i) {
  if (i >= T->size) return -1;
  const node_t *nd = &T->t[i];
  if (nd->a != -1) return (k_le(p, T->pts[nd->a].x) && k_le(T->pts[nd->a].x, q)) ? i : -1;
  const int pv = nd->p;
  if (pv == -1) return -1;
  const key2 px = T->pts[T->t[pv].a].x;
  if (k_le(p, px) && k_le(px, q)) return pv;
  if (k_le(q, T->t[2 * i + 1].h)) return seg_rmq(T, p, q, 2 * i + 1);
  if (k_lt(T->t[2 * i + 1].h, p)) return seg_rmq(T, p, q, 2 * i + 2);
  const int m1 = seg_rmq(T, p, q, 2 * i + 1), m2 = seg_rmq(T, p, q, 2 * i + 2);
  if (m1 == -1) return m2;
  if (m2 == -1) return m1;
  return T->pts[T->t[m1].a].score >= T->pts[T->t[m2].a].score ? m1 : m2; /* :62 the left one on equal scores */
}

static int seg_leaf(const tree_t *T, key2 q) { /* :79-81, :108-110 */
  int leaf = 0;
  while (leaf < T->size && (T->t[leaf].a == -1 || !k_eq(q, T->pts[T->t[leaf].a].x)))
    leaf = 2 * leaf + 1 + (k_lt(T->t[2 * leaf + 1].h, q) ? 1 : 0);
  return leaf;
}

/* src/segment.tpp:76-103: the point sinks from the root; at every node it takes the place of a point that is not better
 * (>=: of an equal one too), which sinks on towards its own leaf */
static void seg_activate(tree_t *T, key2 q, int score) {
  int leaf = seg_leaf(T, q);
  T->pts[T->t[leaf].a].score = score;
  for (int i = 0; i < T->size;) {
    if (T->t[i].p == -1 || T->pts[T->t[leaf].a].score >= T->pts[T->t[T->t[i].p].a].score) {
      const int down = T->t[i].p;
      T->t[i].p = leaf;
      leaf = down;
    }
    if (leaf == -1) break;
    i = 2 * i + 1 + (k_lt(T->t[2 * i + 1].h, T->pts[T->t[leaf].a].x) ? 1 : 0);
  }
}

/* src/segment.tpp:105-146: the node that held the point is refilled from its children -- the right child's point only if
 * it is strictly better than the left child's -- and so on down */
static void seg_deactivate(tree_t *T, key2 q) {
  int leaf = seg_leaf(T, q);
  T->pts[T->t[leaf].a].score = SEG_MIN;
  for (int i = 0; i < T->size;) {
    if (T->t[i].p == -1) break;
    if (T->t[i].p != leaf) {
      i = 2 * i + 1 + (k_lt(T->t[2 * i + 1].h, q) ? 1 : 0);
    } else if (T->t[i].a != -1) {
      T->t[i].p = -1;
    } else {
      const int l = 2 * i + 1, r = 2 * i + 2;
      const int right = r < T->size && T->t[r].p != -1 &&
                        (T->t[l].p == -1 || T->pts[T->t[T->t[r].p].a].score > T->pts[T->t[T->t[l].p].a].score);
      T->t[i].p = leaf = T->t[right ? r : l].p;
      i = right ? r : l;
    }
  }
}

/* Replays a script on the tree; same arguments and results as ref_segtree_script (oracle/ref_align_driver.cc). */
int sdfo_segtree_script(const int *pts_in, int n, const int *ops, int nops, int *out, int *state, int state_cap) {
  if (n < 1) return -1;
  pt_t *pts = (pt_t *)malloc((size_t)n * sizeof(pt_t));
  if (!pts) return -1;
  for (int i = 0; i < n; i++) {
    pts[i].x.a = pts_in[2 * i];
    pts[i].x.b = pts_in[2 * i + 1];
    pts[i].score = SEG_MIN;
    pts[i].pos = i;
  }
  tree_t T;
  if (seg_build(&T, pts, n)) {
    free(pts);
    return -1;
  }
  for (int k = 0; k < nops; k++) {
    const int *o = ops + 5 * k;
    out[2 * k] = out[2 * k + 1] = -2;
    const key2 a = {o[1], o[2]}, b = {o[3], o[4]};
    if (o[0] == 0) {
      seg_activate(&T, a, o[3]);
    } else if (o[0] == 1) {
      seg_deactivate(&T, a);
    } else {
      const int nd = seg_rmq(&T, a, b, 0);
      const int j = nd == -1 ? -1 : T.t[nd].a;
      out[2 * k] = j == -1 ? -1 : pts[j].pos;
      out[2 * k + 1] = j == -1 ? 0 : pts[j].score;
    }
  }
  const int size = T.size;
  if (state)
    for (int i = 0; i < size && i < state_cap; i++) state[i] = T.t[i].p;
  free(T.t);
  free(pts);
  return size;
}

typedef struct {
  key2 x; /* (coordinate, anchor) */
} ev_t;
static int ev_cmp(const void *x, const void *y) {
  const key2 p = ((const ev_t *)x)->x, q = ((const ev_t *)y)->x;
  return k_lt(p, q) ? -1 : k_lt(q, p) ? 1 : 0;
}
typedef struct {
  int score, idx;
} dp_t;
static int dp_desc(const void *x, const void *y) { /* sort(dp, greater<pair<int,int>>), src/chain.cc:179 */
  const dp_t *p = (const dp_t *)x, *q = (const dp_t *)y;
  if (p->score != q->score) return p->score > q->score ? -1 : 1;
  return p->idx > q->idx ? -1 : p->idx < q->idx ? 1 : 0;
}

/* chain_anchors (src/chain.cc:103-199).  anchors: n x {q, r, l, has_u}.  Out: path[n]; bounds[2 (n + 1)] = (path
 * position, any uppercase) pairs, *nbound of them, the first one {0, 0}; dp_out[n], prev_out[n] (may be null).
 * ops_out (may be null, room for 5 * 4n ints): the tree calls of the sweep in ref_segtree_script's format, *nops_out of
 * them -- replayed on the reference's class by the tests. */
int sdfo_chain_anchors(const int *anchors, int n, int max_chain_gap, int match_chain_score, int *path, int *bounds,
                       int *nbound, int *dp_out, int *prev_out, int *ops_out, int *nops_out) {
  bounds[0] = bounds[1] = 0;
  *nbound = 1;
  if (nops_out) *nops_out = 0;
  if (n <= 0) return 0;
  ev_t *xs = (ev_t *)malloc((size_t)2 * n * sizeof(ev_t));
  pt_t *ys = (pt_t *)malloc((size_t)n * sizeof(pt_t));
  int *prev = (int *)malloc((size_t)n * sizeof(int));
  dp_t *dp = (dp_t *)malloc((size_t)n * sizeof(dp_t));
  char *used = (char *)calloc((size_t)n, 1);
  if (!xs || !ys || !prev || !dp || !used) return -1;
  int max_q = 0, max_r = 0, nops = 0;
#define A_Q(i) anchors[4 * (i)]
#define A_R(i) anchors[4 * (i) + 1]
#define A_L(i) anchors[4 * (i) + 2]
#define A_U(i) anchors[4 * (i) + 3]
  for (int i = 0; i < n; i++) { /* :116-126 */
    xs[2 * i].x.a = A_Q(i);
    xs[2 * i].x.b = i;
    xs[2 * i + 1].x.a = A_Q(i) + A_L(i);
    xs[2 * i + 1].x.b = i;
    ys[i].x.a = A_R(i) + A_L(i) - 1;
    ys[i].x.b = i;
    ys[i].score = SEG_MIN;
    ys[i].pos = i;
    if (A_Q(i) + A_L(i) > max_q) max_q = A_Q(i) + A_L(i);
    if (A_R(i) + A_L(i) > max_r) max_r = A_R(i) + A_L(i);
    prev[i] = -1;
    dp[i].score = 0;
    dp[i].idx = i;
  }
  /* (a start and an end event of ONE anchor never compare equal: l > 0) */
  qsort(xs, (size_t)2 * n, sizeof(ev_t), ev_cmp); /* :129 */
  tree_t T;
  if (seg_build(&T, ys, n)) return -1; /* :130 (sorts ys) */
  int bound = 0;
  for (int e = 0; e < 2 * n; e++) { /* :137-178 */
    const int i = xs[e].x.b;
    if (xs[e].x.a == A_Q(i)) { /* a start point */
      while (bound < e) {      /* :141-152: end points too far back in the query leave the tree */
        const int t = xs[bound].x.b;
        if (xs[bound].x.a == A_Q(t) + A_L(t)) {
          if (A_Q(i) - (A_Q(t) + A_L(t)) <= max_chain_gap) break;
          const key2 k = {A_R(t) + A_L(t) - 1, t};
          seg_deactivate(&T, k);
          if (ops_out) {
            int *o = ops_out + 5 * nops++;
            o[0] = 1, o[1] = k.a, o[2] = k.b, o[3] = o[4] = 0;
          }
        }
        bound++;
      }
      const int w = match_chain_score * A_U(i) + (match_chain_score / 2) * (A_L(i) - A_U(i)); /* :155-156 */
      const key2 lo = {A_R(i) - max_chain_gap, 0}, hi = {A_R(i) - 1, n};                       /* :157-158 */
      const int nd = seg_rmq(&T, lo, hi, 0);
      if (ops_out) {
        int *o = ops_out + 5 * nops++;
        o[0] = 2, o[1] = lo.a, o[2] = lo.b, o[3] = hi.a, o[4] = hi.b;
      }
      int j = nd == -1 ? -1 : T.t[nd].a;
      dp[i].score = w;
      if (j != -1 && ys[j].score != SEG_MIN) { /* :159-172 */
        j = ys[j].pos;
        const int gap = A_Q(i) - (A_Q(j) + A_L(j)) + A_R(i) - (A_R(j) + A_L(j));
        if (w + dp[j].score - gap > 0) {
          dp[i].score = w + dp[j].score - gap;
          prev[i] = j;
        }
      }
    } else { /* an end point: the anchor becomes a candidate predecessor (:174-177) */
      const int gap = max_q + 1 - (A_Q(i) + A_L(i)) + max_r + 1 - (A_R(i) + A_L(i));
      const key2 k = {A_R(i) + A_L(i) - 1, i};
      seg_activate(&T, k, dp[i].score - gap);
      if (ops_out) {
        int *o = ops_out + 5 * nops++;
        o[0] = 0, o[1] = k.a, o[2] = k.b, o[3] = dp[i].score - gap, o[4] = 0;
      }
    }
  }
  if (dp_out)
    for (int i = 0; i < n; i++) dp_out[i] = dp[i].score;
  if (prev_out) memcpy(prev_out, prev, (size_t)n * sizeof(int));
  qsort(dp, (size_t)n, sizeof(dp_t), dp_desc); /* :179 */
  int np = 0, nb = 1;
  for (int k = 0; k < n; k++) { /* :185-197 */
    int at = dp[k].idx;
    if (used[at]) continue;
    int has_u = 0;
    while (at != -1 && !used[at]) {
      path[np++] = at;
      has_u += A_U(at);
      used[at] = 1;
      at = prev[at];
    }
    bounds[2 * nb] = np;
    bounds[2 * nb + 1] = has_u != 0; /* vector<pair<int, bool>>: the count narrows to "any" */
    nb++;
  }
  *nbound = nb;
  if (nops_out) *nops_out = nops;
  free(T.t);
  free(xs);
  free(ys);
  free(prev);
  free(dp);
  free(used);
  return 0;
}
