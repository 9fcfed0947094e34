// Anchor chaining on the device: chain_anchors (reference: src/chain.cc:103-199) with the
// priority search tree of src/segment.tpp, one GPU thread per candidate pair.
//
// The sweep over an anchor set is sequential (every anchor's best predecessor depends on the scores of the
// anchors that ended before it), so the parallelism is across the pairs of a super-batch.  The three orderings
// the reference obtains with std::sort have no ties (every key carries the anchor index), so any correct sort
// gives the reference's order; the tree keeps the reference's array layout, split keys and comparison order,
// which is what makes the answer of a range-maximum query unique among equal scores.  Same results as the
// host restatement (sedef_amd/csrc/host/chain.cc), compared in tests/test_host_pipeline.py.
#include <hip/hip_runtime.h>

#include "sdf_internal.h"

namespace sdf {

namespace {

struct P2 {
  int a, b;
};
__device__ __forceinline__ bool lt(const P2 &x, const P2 &y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }
__device__ __forceinline__ bool le(const P2 &x, const P2 &y) { return !lt(y, x); }
__device__ __forceinline__ bool eq(const P2 &x, const P2 &y) { return x.a == y.a && x.b == y.b; }

struct Pt {  // Coor of src/chain.cc:106-110
  P2 x;
  int score, pos;
};
struct Node {  // src/segment.h:21-56
  int p, a;
  P2 h;
};

const int TREE_MIN = (int)0x80000000;

// in-place heap sort; LESS(a, b) is a strict weak order without ties here
template <typename T, typename LESS>
__device__ void heap_sort(T *v, int n, LESS less) {
  auto sift = [&](int root, int end) {
    for (;;) {
      int child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && less(v[child], v[child + 1])) ++child;
      if (!less(v[root], v[child])) break;
      T t = v[root];
      v[root] = v[child];
      v[child] = t;
      root = child;
    }
  };
  for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
  for (int end = n - 1; end > 0; --end) {
    T t = v[0];
    v[0] = v[end];
    v[end] = t;
    sift(0, end);
  }
}

struct Tree {
  Node *tree;
  Pt *pts;
  int size;  // number of nodes

  // src/segment.tpp:172-192, recursion unrolled with an explicit stack (depth <= 32)
  __device__ void build(int npts) {
    int st_i[40], st_s[40], st_e[40], st_state[40];
    int sp = 0, tree_i = 0;
    st_i[0] = 0;
    st_s[0] = 0;
    st_e[0] = npts;
    st_state[0] = 0;
    while (sp >= 0) {
      const int i = st_i[sp], s = st_s[sp], e = st_e[sp];
      if (i >= size) {
        --sp;
        continue;
      }
      if (st_state[sp] == 0) {
        if (s + 1 == e) {
          tree[i].p = -1;
          tree[i].a = tree_i;
          tree[i].h = pts[tree_i].x;
          pts[tree_i].score = TREE_MIN;
          ++tree_i;
          --sp;
          continue;
        }
        const int bnd = (s + e + 1) / 2;
        st_state[sp] = 1;
        ++sp;
        st_i[sp] = 2 * i + 1;
        st_s[sp] = s;
        st_e[sp] = bnd;
        st_state[sp] = 0;
      } else if (st_state[sp] == 1) {
        const int bnd = (s + e + 1) / 2;
        st_state[sp] = 2;
        ++sp;
        st_i[sp] = 2 * i + 2;
        st_s[sp] = bnd;
        st_e[sp] = e;
        st_state[sp] = 0;
      } else {
        tree[i].p = -1;
        tree[i].a = -1;
        tree[i].h = tree[2 * i + 1 + (2 * i + 2 < size)].h;
        --sp;
      }
    }
  }

  // src/segment.tpp:29-66; returns a node index or -1
  __device__ int rmq(const P2 &p, const P2 &q) const {
    int st_i[48], st_state[48], st_m1[48];  // (dynamically indexed: scratch memory)
    return rmq(p, q, st_i, st_state, st_m1);
  }
  // ... with the recursion's stack where the caller wants it (chain_wave_kernel: in LDS, 3 x 48 words)
  __device__ int rmq(const P2 &p, const P2 &q, int *st_i, int *st_state, int *st_m1) const {
    int sp = 0, ret = -1;
    st_i[0] = 0;
    st_state[0] = 0;
    while (sp >= 0) {
      int i = st_i[sp];
      if (st_state[sp] == 0) {
        // descend along single-child cases
        for (;;) {
          if (i >= size) {
            ret = -1;
            break;
          }
          if (tree[i].a != -1) {
            ret = (le(p, pts[tree[i].a].x) && le(pts[tree[i].a].x, q)) ? i : -1;
            break;
          }
          const int pv = tree[i].p;
          if (pv == -1) {
            ret = -1;
            break;
          }
          if (le(p, pts[tree[pv].a].x) && le(pts[tree[pv].a].x, q)) {
            ret = pv;
            break;
          }
          if (le(q, tree[2 * i + 1].h)) {
            i = 2 * i + 1;
            continue;
          }
          if (lt(tree[2 * i + 1].h, p)) {
            i = 2 * i + 2;
            continue;
          }
          // both children: left first
          st_i[sp] = i;
          st_state[sp] = 1;
          ++sp;
          st_i[sp] = 2 * i + 1;
          st_state[sp] = 0;
          ret = -2;  // marker: pushed
          break;
        }
        if (ret != -2) --sp;  // this frame is done, `ret` holds its value
      } else if (st_state[sp] == 1) {
        st_m1[sp] = ret;
        st_state[sp] = 2;
        ++sp;
        st_i[sp] = 2 * i + 2;
        st_state[sp] = 0;
      } else {
        const int m1 = st_m1[sp], m2 = ret;
        if (m1 == -1) ret = m2;
        else if (m2 == -1) ret = m1;
        else ret = pts[tree[m1].a].score >= pts[tree[m2].a].score ? m1 : m2;
        --sp;
      }
    }
    return ret;
  }

  __device__ int find_leaf(const P2 &q) const {
    int leaf = 0;
    while (leaf < size && (tree[leaf].a == -1 || !eq(q, pts[tree[leaf].a].x)))
      leaf = 2 * leaf + 1 + (lt(tree[2 * leaf + 1].h, q) ? 1 : 0);
    return leaf;
  }

  __device__ void activate(const P2 &q, int score) {  // src/segment.tpp:76-103
    int leaf = find_leaf(q);
    pts[tree[leaf].a].score = score;
    for (int i = 0; i < size;) {
      if (tree[i].p == -1 || pts[tree[leaf].a].score >= pts[tree[tree[i].p].a].score) {
        const int t = tree[i].p;
        tree[i].p = leaf;
        leaf = t;
      }
      if (leaf == -1) break;
      i = 2 * i + 1 + (lt(tree[2 * i + 1].h, pts[tree[leaf].a].x) ? 1 : 0);
    }
  }

  __device__ void deactivate(const P2 &q) {  // src/segment.tpp:105-146
    int leaf = find_leaf(q);
    pts[tree[leaf].a].score = TREE_MIN;
    for (int i = 0; i < size;) {
      if (tree[i].p == -1) break;
      if (tree[i].p == leaf) {
        if (tree[i].a != -1) {
          tree[i].p = -1;
        } else if (2 * i + 2 < size && tree[2 * i + 2].p != -1 &&
                   (tree[2 * i + 1].p == -1 ||
                    pts[tree[tree[2 * i + 2].p].a].score > pts[tree[tree[2 * i + 1].p].a].score)) {
          tree[i].p = leaf = tree[2 * i + 2].p;
          i = 2 * i + 2;
        } else {
          tree[i].p = leaf = tree[2 * i + 1].p;
          i = 2 * i + 1;
        }
      } else {
        i = 2 * i + 1 + (lt(tree[2 * i + 1].h, q) ? 1 : 0);
      }
    }
  }
};

}  // namespace

// One thread per pair.  anchors[off[p] .. off[p+1]) in the order of generate_anchors; work: per-pair scratch at
// ws_off[p] (in 32-bit words: 12 m + 4 nodes).  Out: path[off[p] + k] = anchor index (within the pair) of the k-th
// element of the reference's `path`; bounds[2 * (off[p] + p + b)] = {path position, has_u} of the b-th boundary;
// nbound[p] = number of boundaries (>= 1: the initial {0, 0}).
// (which: the pairs this launch takes, `npairs` of them -- round 4: those too large for the LDS of chain_wave_kernel; null: all)
__global__ __launch_bounds__(64) void chain_kernel(const sdf_anchor *__restrict__ anchors,
                                                   const int64_t *__restrict__ off, const int64_t *__restrict__ ws_off,
                                                   int npairs, int max_chain_gap, int match_chain_score,
                                                   int32_t *__restrict__ work, int32_t *__restrict__ path,
                                                   int32_t *__restrict__ bounds, int32_t *__restrict__ nbound,
                                                   const int32_t *__restrict__ which) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= npairs) return;
  const int p = which ? which[slot] : slot;
  const sdf_anchor *A = anchors + off[p];
  const int m = (int)(off[p + 1] - off[p]);
  int32_t *pth = path + off[p];
  int32_t *bnd = bounds + 2 * (off[p] + p);
  bnd[0] = 0;
  bnd[1] = 0;
  int nb = 1;
  if (m == 0) {
    nbound[p] = nb;
    return;
  }
  int bits = 0;
  for (unsigned v = (unsigned)m - 1u; v; v >>= 1) ++bits;
  const int tsize = (1 << bits) << 1;
  int32_t *w = work + ws_off[p];
  P2 *xs = reinterpret_cast<P2 *>(w);                      // 2m events (x, anchor)
  Pt *ys = reinterpret_cast<Pt *>(w + 4 * m);              // m points
  Node *nodes = reinterpret_cast<Node *>(w + 8 * m);       // tsize nodes
  int32_t *prev = w + 8 * m + 4 * tsize;                   // m
  P2 *dp = reinterpret_cast<P2 *>(w + 9 * m + 4 * tsize);  // m (score, anchor)
  int32_t *used = w + 11 * m + 4 * tsize;                  // m

  int max_q = 0, max_r = 0;
  for (int i = 0; i < m; ++i) {
    const sdf_anchor a = A[i];
    xs[2 * i] = P2{a.q, i};
    xs[2 * i + 1] = P2{a.q + a.l, i};
    ys[i] = Pt{P2{a.r + a.l - 1, i}, TREE_MIN, i};
    max_q = max(max_q, a.q + a.l);
    max_r = max(max_r, a.r + a.l);
    prev[i] = -1;
    dp[i] = P2{0, i};
    used[i] = 0;
  }
  heap_sort(xs, 2 * m, [](const P2 &x, const P2 &y) { return lt(x, y); });
  heap_sort(ys, m, [](const Pt &x, const Pt &y) { return lt(x.x, y.x); });
  Tree tr{nodes, ys, tsize};
  tr.build(m);

  int deactivate_bound = 0;
  for (int xi = 0; xi < 2 * m; ++xi) {
    const int i = xs[xi].b;
    const sdf_anchor a = A[i];
    if (xs[xi].a == a.q) {  // start point
      while (deactivate_bound < xi) {
        const int t = xs[deactivate_bound].b;
        if (xs[deactivate_bound].a == A[t].q + A[t].l) {  // an end point
          if (a.q - (A[t].q + A[t].l) <= max_chain_gap) break;
          tr.deactivate(P2{A[t].r + A[t].l - 1, t});
        }
        ++deactivate_bound;
      }
      const int wgt = match_chain_score * a.has_u + (match_chain_score / 2) * (a.l - a.has_u);
      const int node = tr.rmq(P2{a.r - max_chain_gap, 0}, P2{a.r - 1, m});
      int j = node == -1 ? -1 : nodes[node].a;
      if (j != -1 && ys[j].score != TREE_MIN) {
        j = ys[j].pos;
        const sdf_anchor pa = A[j];
        const int gap = (a.q - (pa.q + pa.l) + a.r - (pa.r + pa.l));
        if (wgt + dp[j].a - gap > 0) {
          dp[i].a = wgt + dp[j].a - gap;
          prev[i] = j;
        } else {
          dp[i].a = wgt;
        }
      } else {
        dp[i].a = wgt;
      }
    } else {  // end point: the anchor becomes available as a predecessor
      const int gap = (max_q + 1 - (a.q + a.l) + max_r + 1 - (a.r + a.l));
      tr.activate(P2{a.r + a.l - 1, i}, dp[i].a - gap);
    }
  }
  // NB: dp[] is indexed by anchor above and sorted (descending) only now
  heap_sort(dp, m, [](const P2 &x, const P2 &y) { return lt(y, x); });
  int np = 0;
  for (int k = 0; k < m; ++k) {
    int maxi = dp[k].b;
    if (used[maxi]) continue;
    int has_u = 0;
    while (maxi != -1 && !used[maxi]) {
      pth[np++] = maxi;
      has_u += A[maxi].has_u;
      used[maxi] = 1;
      maxi = prev[maxi];
    }
    bnd[2 * nb] = np;
    bnd[2 * nb + 1] = has_u != 0;  // int -> bool: "any uppercase anchor"
    ++nb;
  }
  nbound[p] = nb;
}

// ---- round 4: one WAVEFRONT per pair, everything in LDS ----------------------------------------------------------------
// The sweep itself stays what it is -- a chain of tree operations whose tie winners depend on the tree's shape and history
// (DESIGN.md 8, f3), walked by lane 0 with the device functions above -- but (1) every array it touches lies in LDS (a
// dependent access costs ~100 cycles instead of the ~1,500 of a miss in HBM: the thread-per-pair kernel spends ~100 us per
// anchor), and (2) what is not a chain runs on all 64 lanes: the three sorts (a bitonic network on 64-bit keys, every
// compare-exchange ascending so that lengths need not be powers of two), the initialisation, and the tree's construction
// (a node's point range follows from its index alone).  LDS per pair: 64 m + 16 nodes bytes (m anchors -- their records
// included --, nodes = 2 * 2^ceil(log2 m) <= 4 m); pairs beyond the launch's LDS go to the thread-per-pair kernel.

// ascending sort of a[0 .. n) in LDS by the workgroup's single wavefront (normalised bitonic network: the first step of a
// merge pairs i with its mirror image in the block, the others i with i + j; a partner at or beyond n is a virtual +inf)
__device__ __forceinline__ void chain_sort_u64(unsigned long long *a, const int n, const int lane) {
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int k = 2; k <= np2; k <<= 1) {
    for (int idx = lane; idx < np2 / 2; idx += 64) {
      const int blk = idx / (k / 2), off = idx % (k / 2);
      const int i = blk * k + off, j = blk * k + k - 1 - off;
      if (j < n) {
        const unsigned long long x = a[i], y = a[j];
        if (x > y) a[i] = y, a[j] = x;
      }
    }
    __syncthreads();
    for (int jj = k / 4; jj >= 1; jj >>= 1) {
      for (int idx = lane; idx < np2 / 2; idx += 64) {
        const int i = (idx / jj) * 2 * jj + idx % jj, j = i + jj;
        if (j < n) {
          const unsigned long long x = a[i], y = a[j];
          if (x > y) a[i] = y, a[j] = x;
        }
      }
      __syncthreads();
    }
  }
}

__host__ __device__ inline size_t chain_wave_lds_bytes(int m) {
  if (m <= 0) return 16;
  int bits = 0;
  for (unsigned v = (unsigned)m - 1u; v; v >>= 1) ++bits;
  return (size_t)64 * m + (size_t)16 * ((size_t)2 << bits) + 3 * 48 * 4 + 64;
}

// grid: one workgroup of 64 lanes per entry of `which` (pair indices, those whose arrays fit `lds_cap`)
__global__ __launch_bounds__(64) void chain_wave_kernel(const sdf_anchor *__restrict__ anchors, const int64_t *__restrict__ off,
                                                        const int32_t *__restrict__ which, int max_chain_gap, int match_chain_score,
                                                        int32_t *__restrict__ path, int32_t *__restrict__ bounds,
                                                        int32_t *__restrict__ nbound) {
  extern __shared__ __align__(16) unsigned char chain_lds[];
  const int p = which[blockIdx.x], lane = threadIdx.x;
  const sdf_anchor *A = anchors + off[p];
  const int m = (int)(off[p + 1] - off[p]);
  int32_t *pth = path + off[p];
  int32_t *bnd = bounds + 2 * (off[p] + p);
  if (m == 0) {
    if (lane == 0) bnd[0] = 0, bnd[1] = 0, nbound[p] = 1;
    return;
  }
  int bits = 0;
  for (unsigned v = (unsigned)m - 1u; v; v >>= 1) ++bits;
  const int tsize = (1 << bits) << 1;
  // layout: kx [2m] u64 (event keys; after the sweep: the dp keys) | ys [m] Pt | nodes [tsize] Node | anchors [m] | prev [m] |
  // used [m] | dpv [m]  (the keys of the points, ky, lie over prev + used until the points are written)
  unsigned long long *kx = reinterpret_cast<unsigned long long *>(chain_lds);
  Pt *ys = reinterpret_cast<Pt *>(kx + 2 * m);
  Node *nodes = reinterpret_cast<Node *>(ys + m);
  sdf_anchor *L = reinterpret_cast<sdf_anchor *>(nodes + tsize);
  int32_t *prev = reinterpret_cast<int32_t *>(L + m), *used = prev + m, *dpv = used + m;
  int32_t *stk = dpv + m;  // the range query's recursion stack: 3 x 48 words (private arrays would be scratch memory in HBM)
  unsigned long long *ky = reinterpret_cast<unsigned long long *>(prev);
  static_assert(sizeof(sdf_anchor) == 16 && sizeof(Pt) == 16 && sizeof(Node) == 16, "sixteen-byte records");
  int max_q = 0, max_r = 0;
  for (int i = lane; i < m; i += 64) {
    const sdf_anchor a = A[i];
    L[i] = a;
    kx[2 * i] = ((unsigned long long)(unsigned)a.q << 32) | (unsigned)i;            // start event (x, anchor)
    kx[2 * i + 1] = ((unsigned long long)(unsigned)(a.q + a.l) << 32) | (unsigned)i;  // end event
    ky[i] = ((unsigned long long)(unsigned)(a.r + a.l - 1) << 32) | (unsigned)i;
    max_q = max(max_q, a.q + a.l);
    max_r = max(max_r, a.r + a.l);
    dpv[i] = 0;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    max_q = max(max_q, __shfl_xor(max_q, o));
    max_r = max(max_r, __shfl_xor(max_r, o));
  }
  __syncthreads();
  chain_sort_u64(kx, 2 * m, lane);
  chain_sort_u64(ky, m, lane);
  // the points in their sorted order, then (ky is dead) prev / used
  for (int i = lane; i < m; i += 64) {
    const unsigned long long kv = ky[i];
    ys[i] = Pt{P2{(int)(kv >> 32), (int)(kv & 0xffffffffu)}, TREE_MIN, (int)(kv & 0xffffffffu)};
  }
  __syncthreads();
  for (int i = lane; i < m; i += 64) prev[i] = -1, used[i] = 0;
  // the tree (src/segment.tpp:172-192): node i covers the point range [s, e) its index implies -- root [0, m), children
  // [s, (s + e + 1) / 2) and [(s + e + 1) / 2, e) --; one point: a leaf holding point s; more: h = the key of its last point
  for (int i = lane; i < tsize; i += 64) {
    int depth = 0;
    for (unsigned v = (unsigned)i + 1u; v > 1u; v >>= 1) ++depth;
    int sgm = 0, e = m;
    bool exists = true;
    for (int d = depth - 1; d >= 0 && exists; --d) {
      if (sgm + 1 >= e) {
        exists = false;  // an ancestor is already a leaf
        break;
      }
      const int bndp = (sgm + e + 1) / 2;
      if ((((unsigned)i + 1u) >> d) & 1u) sgm = bndp;  // (bit d of i + 1 below its leading one: 1 = right child)
      else e = bndp;
    }
    Node nd;
    nd.p = -1;
    nd.a = -1;
    nd.h = P2{0, 0};
    if (exists && sgm < e) {
      nd.a = sgm + 1 == e ? sgm : -1;
      nd.h = ys[e - 1].x;
    }
    nodes[i] = nd;
  }
  __syncthreads();
  if (lane == 0) {
    Tree tr{nodes, ys, tsize};
    int deactivate_bound = 0;
    for (int xi = 0; xi < 2 * m; ++xi) {
      const unsigned long long ev = kx[xi];
      const int i = (int)(ev & 0xffffffffu), x = (int)(ev >> 32);
      const sdf_anchor a = L[i];
      if (x == a.q) {  // start point
        while (deactivate_bound < xi) {
          const unsigned long long dv = kx[deactivate_bound];
          const int t = (int)(dv & 0xffffffffu);
          const sdf_anchor at = L[t];
          if ((int)(dv >> 32) == at.q + at.l) {  // an end point
            if (a.q - (at.q + at.l) <= max_chain_gap) break;
            tr.deactivate(P2{at.r + at.l - 1, t});
          }
          ++deactivate_bound;
        }
        const int wgt = match_chain_score * a.has_u + (match_chain_score / 2) * (a.l - a.has_u);
        const int node = tr.rmq(P2{a.r - max_chain_gap, 0}, P2{a.r - 1, m}, stk, stk + 48, stk + 96);
        int j = node == -1 ? -1 : nodes[node].a;
        int val = wgt;
        if (j != -1 && ys[j].score != TREE_MIN) {
          j = ys[j].pos;
          const sdf_anchor pa = L[j];
          const int gap = (a.q - (pa.q + pa.l) + a.r - (pa.r + pa.l));
          if (wgt + dpv[j] - gap > 0) {
            val = wgt + dpv[j] - gap;
            prev[i] = j;
          }
        }
        dpv[i] = val;
      } else {  // end point: the anchor becomes available as a predecessor
        const int gap = (max_q + 1 - (a.q + a.l) + max_r + 1 - (a.r + a.l));
        tr.activate(P2{a.r + a.l - 1, i}, dpv[i] - gap);
      }
    }
  }
  __syncthreads();
  // sort(dp, greater) on (score, anchor): ascending on the complemented key
  for (int i = lane; i < m; i += 64) kx[i] = ~(((unsigned long long)(unsigned)dpv[i] << 32) | (unsigned)i);
  __syncthreads();
  chain_sort_u64(kx, m, lane);
  if (lane == 0) {
    bnd[0] = 0;
    bnd[1] = 0;
    int nb = 1, np = 0;
    for (int k = 0; k < m; ++k) {
      int maxi = (int)(~kx[k] & 0xffffffffu);
      if (used[maxi]) continue;
      int has_u = 0;
      while (maxi != -1 && !used[maxi]) {
        pth[np++] = maxi;
        has_u += L[maxi].has_u;
        used[maxi] = 1;
        maxi = prev[maxi];
      }
      bnd[2 * nb] = np;
      bnd[2 * nb + 1] = has_u != 0;  // int -> bool: "any uppercase anchor"
      ++nb;
    }
    nbound[p] = nb;
  }
}

// Test hook (tests/test_chain_oracle.py): one thread replays a script of activate / deactivate / rmq calls on the device
// tree above, in the script format of oracle/ref_align_driver.cc: ref_segtree_script (which drives the reference's own
// SegmentTree class).  work: 4 n + 4 size words (points, nodes).
__global__ void chain_tree_script_kernel(const int32_t *__restrict__ pts_in, int n, const int32_t *__restrict__ ops, int nops,
                                         int32_t *__restrict__ work, int size, int32_t *__restrict__ out,
                                         int32_t *__restrict__ state) {
  if (blockIdx.x || threadIdx.x) return;
  Pt *pts = reinterpret_cast<Pt *>(work);
  Node *nodes = reinterpret_cast<Node *>(work + 4 * n);
  for (int i = 0; i < n; ++i) pts[i] = Pt{P2{pts_in[2 * i], pts_in[2 * i + 1]}, TREE_MIN, i};
  heap_sort(pts, n, [](const Pt &x, const Pt &y) { return lt(x.x, y.x); });
  for (int i = 0; i < size; ++i) nodes[i] = Node{-1, -1, P2{0, 0}};
  Tree tr{nodes, pts, size};
  tr.build(n);
  for (int k = 0; k < nops; ++k) {
    const int32_t *o = ops + 5 * k;
    out[2 * k] = out[2 * k + 1] = -2;
    if (o[0] == 0) {
      tr.activate(P2{o[1], o[2]}, o[3]);
    } else if (o[0] == 1) {
      tr.deactivate(P2{o[1], o[2]});
    } else {
      const int nd = tr.rmq(P2{o[1], o[2]}, P2{o[3], o[4]});
      const int j = nd == -1 ? -1 : nodes[nd].a;
      out[2 * k] = j == -1 ? -1 : pts[j].pos;
      out[2 * k + 1] = j == -1 ? 0 : pts[j].score;
    }
  }
  for (int i = 0; i < size; ++i) state[i] = nodes[i].p;
}

}  // namespace sdf
