// GPU seed anchors: maximal exact k-mer matches between the two sequences of a candidate pair.
//
// Restates generate_anchors (reference: src/chain.cc:24-101) as sort / search / scan kernels and returns the
// anchors in the reference's order (query position ascending, then reference position ascending):
//   1. every reference k-mer (2 bits per base, windows with an N skipped, :28-40) becomes a 64-bit key
//      pair | hash (2k bits) | position (as many bits as the call's longest reference needs); one radix sort over the
//      bits in use groups equal k-mers of a pair, positions ascending.  A call whose pairs do not fit the bits left of
//      hash and position (k up to 15, references up to 2 Gb) is run range of pairs by range of pairs (sdf_api.hip);
//   2. every query k-mer finds its run of equal keys by binary search; runs of >= 1000 are disabled (:61);
//   3. an exclusive scan of the run lengths enumerates the candidate (q, r) pairs in reference order;
//   4. a candidate starts an anchor iff no earlier enabled k-mer of the same uninterrupted match run lies on its
//      diagonal -- that is what the per-diagonal `slide[]` bookkeeping (:42,70-72,92) amounts to, because an
//      anchor always extends to the end of its run; starts are extended to the right (:76-85) and compacted.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "sdf_internal.h"

namespace sdf {

struct AnchorPairDev {
  int64_t q_off, r_off;    // byte offsets of the raw sequences in the pool
  int32_t qlen, rlen;
  int32_t same_chr, delta;  // near-diagonal filter of self comparisons (:67-69)
  int64_t rk_start, qk_start;  // first global k-mer index of this pair's reference / query
};

__device__ __forceinline__ int up(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
__device__ __forceinline__ bool is_upper(int c) { return c >= 'A' && c <= 'Z'; }
__device__ __forceinline__ int base2(int c) {  // hash_dna (src/common.h:69,89): ACGT -> 0..3, anything else 0
  c = up(c);
  return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
}
// hash of the k-mer starting at s; returns false if the window holds an N
__device__ __forceinline__ bool kmer_at(const char *s, int k, uint32_t &h) {
  h = 0;
  bool ok = true;
  for (int i = 0; i < k; i++) {
    const int c = s[i];
    ok = ok && up(c) != 'N';
    h = (h << 2) | (uint32_t)base2(c);
  }
  return ok;
}

// (grid: x = 32 blocks over a pair's k-mers, y x z = the pairs, 65,535 rows per layer; pos_bits: bits of the position field)
__global__ __launch_bounds__(256) void ref_keys_kernel(const AnchorPairDev *pairs, int npairs, const char *pool, int k,
                                                       int pos_bits, unsigned long long *keys) {
  const unsigned pr = blockIdx.z * 65535u + blockIdx.y;
  if (pr >= (unsigned)npairs) return;
  const AnchorPairDev p = pairs[pr];
  const int nk = p.rlen - k + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += gridDim.x * blockDim.x) {
    uint32_t h;
    const bool ok = kmer_at(pool + p.r_off + i, k, h);
    keys[p.rk_start + i] = ok ? (((unsigned long long)pr << (2 * k + pos_bits)) | ((unsigned long long)h << pos_bits) | (unsigned)i)
                              : ~0ull;
  }
}

__device__ __forceinline__ long long lower_bound_u64(const unsigned long long *a, long long n, unsigned long long v) {
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void query_lookup_kernel(const AnchorPairDev *pairs, int npairs, const char *pool, int k,
                                                           int pos_bits, const unsigned long long *keys, long long nkeys,
                                                           uint32_t *qlo, uint32_t *qcnt, uint32_t *qeff,
                                                           uint32_t *qpair) {
  const unsigned pr = blockIdx.z * 65535u + blockIdx.y;
  if (pr >= (unsigned)npairs) return;
  const AnchorPairDev p = pairs[pr];
  const int nk = p.qlen - k + 1;
  // The pair's keys are one run of the sorted array (the keys of windows with an N, all ones, have gathered at the array's
  // end: a pair's keys no longer lie where they were written): one thread finds the run, every k-mer is then looked up inside
  // it -- 17 probes into 800 kB for a 100 kb reference instead of 27 into the whole array -- and the end of its matches by
  // galloping from their start (most k-mers have none or a few).  Round 4: 12.2 -> see profiles/r04_stage.txt.
  __shared__ long long seg[2];
  if (threadIdx.x == 0) {
    const int shift = 2 * k + pos_bits;
    const unsigned long long first = (unsigned long long)pr << shift, next = ((unsigned long long)pr + 1) << shift;
    seg[0] = lower_bound_u64(keys, nkeys, first);
    seg[1] = lower_bound_u64(keys, nkeys, next > first ? next : ~0ull);  // (the last pair of a 64-bit key: up to the all-ones keys)
  }
  __syncthreads();
  const long long s_lo = seg[0], s_hi = seg[1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += gridDim.x * blockDim.x) {
    uint32_t h;
    const bool ok = kmer_at(pool + p.q_off + i, k, h);
    uint32_t lo = 0, cnt = 0;
    if (ok) {
      const unsigned long long base = ((unsigned long long)pr << (2 * k + pos_bits)) | ((unsigned long long)h << pos_bits);
      const unsigned long long top0 = base + (1ull << pos_bits);  // (wraps for the all-T k-mer of the last pair of a 64-bit key)
      const unsigned long long top = top0 > base ? top0 : ~0ull;
      const long long a = s_lo + lower_bound_u64(keys + s_lo, s_hi - s_lo, base);
      long long b = a;
      if (a < s_hi && keys[a] < top) {
        long long at = a, step = 1;  // keys[at] < top
        while (at + step < s_hi && keys[at + step] < top) {
          at += step;
          step <<= 1;
        }
        const long long hi = at + step < s_hi ? at + step : s_hi;  // keys[hi] >= top, or the run's end
        b = at + 1 + lower_bound_u64(keys + at + 1, hi - (at + 1), top);
      }
      lo = (uint32_t)a;
      cnt = (uint32_t)(b - a);
    }
    const long long g = p.qk_start + i;
    qlo[g] = lo;
    qcnt[g] = cnt;
    qeff[g] = cnt < 1000 ? cnt : 0;  // it->second.size() >= 1000 -> skipped (:61)
    qpair[g] = pr;
  }
}

struct CandOut {
  int32_t q, r, l, has_u;
};

__global__ __launch_bounds__(256) void candidates_kernel(const AnchorPairDev *pairs, const char *pool, int k,
                                                         const unsigned long long *keys, const uint32_t *qlo,
                                                         const uint32_t *qcnt, const unsigned long long *cand_off,
                                                         const uint32_t *qpair, long long nq, long long ncand,
                                                         uint32_t *flag, CandOut *cand, int pos_bits) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand) return;
  // owner query k-mer: last g with cand_off[g] <= c
  long long lo = 0, hi = nq;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (cand_off[mid] <= (unsigned long long)c) lo = mid + 1; else hi = mid;
  }
  const long long g = lo - 1;
  const AnchorPairDev p = pairs[qpair[g]];
  const int q = (int)(g - p.qk_start);
  const int j = (int)((unsigned long long)c - cand_off[g]);
  const int r = (int)(keys[qlo[g] + j] & ((1ull << pos_bits) - 1));
  const char *Q = pool + p.q_off, *R = pool + p.r_off;
  bool start = !(p.same_chr && abs(p.delta + r - q) <= k);
  // an earlier enabled k-mer of the same match run on this diagonal already covers this one
  for (int s = 1; start; s++) {
    const int qq = q - s, rr = r - s;
    if (qq < 0 || rr < 0) break;
    const int cq = up(Q[qq]), cr = up(R[rr]);
    if (cq == 'N' || cr == 'N' || cq != cr) break;
    if (qcnt[g - s] < 1000) start = false;  // enabled: its anchor (or an even earlier one) extends over q
  }
  uint32_t f = 0;
  if (start) {
    int len = 0;
    bool hu = false;
    for (; q + len < p.qlen && r + len < p.rlen; len++) {
      const int a = Q[q + len], b = R[r + len];
      if (up(a) == 'N' || up(b) == 'N' || up(a) != up(b)) break;
      hu = hu || is_upper(a) || is_upper(b);
    }
    if (len >= k) {
      f = 1;
      cand[c] = CandOut{q, r, len, hu ? 1 : 0};
    }
  }
  flag[c] = f;
}

__global__ __launch_bounds__(256) void anchors_compact_kernel(const uint32_t *flag, const unsigned long long *pos,
                                                              const CandOut *cand, long long ncand, CandOut *out,
                                                              unsigned long long cap) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncand || !flag[c]) return;
  if (pos[c] < cap) out[pos[c]] = cand[c];
}

// anchors before pair p = kept candidates before the first candidate of the pair's first query k-mer
__global__ void anchor_offsets_kernel(const AnchorPairDev *pairs, int npairs, const unsigned long long *cand_off,
                                      const unsigned long long *pos, long long ncand, unsigned long long total,
                                      long long nq, long long *out_off) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > npairs) return;
  if (p == npairs) {
    out_off[p] = (long long)total;
    return;
  }
  const long long g = pairs[p].qk_start;
  const unsigned long long c0 = g < nq ? cand_off[g] : (unsigned long long)ncand;
  out_off[p] = c0 < (unsigned long long)ncand ? (long long)pos[c0] : (long long)total;
}

}  // namespace sdf
