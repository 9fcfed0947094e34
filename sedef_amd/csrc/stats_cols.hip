// Per-alignment columns of `sedef stats generate` (scope row f4) for a batch of finished alignments.
//
// Reference: process() walks the three column strings that populate_nice_alignment expanded from the CIGAR
// (src/align.cc:274-315) and counts, per column, indels, matches, mismatches split into transitions and
// transversions, and upper-case (not soft-masked) bases (src/stats_main.cc:228-270); the AlignmentError counters
// {gaps, gap_bases, mismatches, matches} come from populate_nice_alignment itself (src/align.cc:300-314, ceq :29-35).
// Here nothing is expanded: one wavefront takes one alignment, 64 CIGAR runs at a time.  The runs' unit / a / b offsets
// are wave scans; the chunk is cut into units of up to eight consecutive columns of one run, every lane takes every 64th
// unit (two per round, so that four loads are in flight), finds its run by a six-step search of the chunk's unit offsets
// in LDS, reads its characters with one unaligned 8-byte load per sequence and counts them four columns per 32-bit word.
// HBM-bound byte pass: a_len + b_len + 4 n_cigar + 64 bytes per alignment.
//
// A long alignment (more than STATS_LONG runs) would be one wavefront's work for milliseconds: its wavefront only sums the
// advances of its runs, STATS_SEG at a time, and leaves one work item per segment -- an alignment of its own: the sequences'
// ranges the segment's runs consume -- in a list (stats_columns_kernel); a second launch of a fixed number of wavefronts
// takes the items and adds their counters to the alignment's record (stats_segments_kernel).
#pragma once
#include "sdf_internal.h"

namespace sdf {

constexpr int STATS_WAVES = 4;  // alignments per workgroup
constexpr uint32_t STATS_SEG = 512, STATS_LONG = 1024;  // runs per segment of a long alignment / runs that make one long

struct StatsItem {  // a segment of a long alignment
  sdf_stats_task t;
  uint32_t task;  // the alignment it belongs to (0xffffffff: nothing to do)
  uint32_t pad;
};

// Prefix sums and sums over the wavefront (all 64 lanes active): an inclusive scan inside each row of sixteen lanes
// (row_shr 1, 2, 4, 8), the rows' totals passed on (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3) --
// six DPP adds, no LDS; the sum is lane 63's prefix.
__device__ __forceinline__ int stats_wave_scan(int v) {  // inclusive prefix sum over the lanes
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ int stats_wave_sum(int v) { return __builtin_amdgcn_readlane(stats_wave_scan(v), 63); }

// Eight consecutive characters of a sequence, as many of them as the sequence still holds (the rest unspecified):
// one unaligned 8-byte load, taken from the last eight bytes of the sequence when fewer remain.
__device__ __forceinline__ uint64_t stats_ld8(const char *p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ uint64_t stats_fetch8(const char *s, int pos, int slen, bool wide) {
  const int avail = slen - pos;
  if (wide) {  // wave-uniform: the sequence holds eight bytes
    const bool whole = avail >= 8;
    const uint64_t v = stats_ld8(s + (whole ? pos : slen - 8));
    return whole ? v : v >> (8 * (8 - avail));
  }
  uint64_t v = 0;
  for (int i = 0; i < avail && i < 8; i++) v |= (uint64_t)(unsigned char)s[pos + i] << (8 * i);
  return v;
}

constexpr uint64_t STATS_DASHES = 0x2D2D2D2D2D2D2D2DULL;

// the counters of one alignment (or segment), summed over the wavefront: v[0..11], and whether its CIGAR fits
__device__ __forceinline__ int stats_count_alignment(const sdf_stats_task &T, const char *__restrict__ pool,
                                                     const uint32_t *__restrict__ cigars, int *unit, int *sa, int *sb, int *sl,
                                                     const int lane, int (&v)[12]) {
  const char *a = pool + T.a_off, *b = pool + T.b_off;
  const uint32_t *cg = cigars + T.cigar_off;
  const int n_cigar = (int)T.n_cigar, a_len = (int)T.a_len, b_len = (int)T.b_len;
  const bool wide_a = a_len >= 8, wide_b = b_len >= 8;

  // mismatchB = alnB - matchB, transversionsB = mismatchB - transitionsB, mismatches = alnB - matches: derived at the end
  int indel_a = 0, indel_b = 0, aln_b = 0, match_b = 0, ts = 0, up_a = 0, up_b = 0, up_m = 0;
  int matches = 0, gaps = 0, gap_bases = 0;
  int ia = 0, ib = 0, bad = 0;  // wave-uniform
  int span_l = 0;

  // a unit: up to eight consecutive columns of one run.  fetch() finds unit u's run by a six-step search of the
  // chunk's unit offsets and loads its characters; lanes past the last unit get eight ('-', '-') columns.
  auto fetch = [&](int u, int total, uint64_t &wa, uint64_t &wb, int &cnt) {
    const bool valid = u < total;
    u = valid ? u : total - 1;
    int j = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1)
      if (unit[j + step] <= u) j += step;  // last run that starts at or before unit u: the one that holds it
    const int d = 8 * (u - unit[j]), pa = sa[j], pb = sb[j];
    const int left = sl[j] - d;
    cnt = valid ? (left < 8 ? left : 8) : 0;
    wa = pa >= 0 ? stats_fetch8(a, pa + d, a_len, wide_a) : STATS_DASHES;
    wb = pb >= 0 ? stats_fetch8(b, pb + d, b_len, wide_b) : STATS_DASHES;
    const uint64_t keep = cnt >= 8 ? ~0ULL : (1ULL << (8 * cnt)) - 1ULL;
    wa = (wa & keep) | (STATS_DASHES & ~keep);
    wb = (wb & keep) | (STATS_DASHES & ~keep);
  };
  // One column at a time: the statement of the counters (src/stats_main.cc:239-269), used for units that hold bytes
  // outside ASCII; everything else goes through count() below.
  auto count_scalar = [&](uint64_t wa, uint64_t wb) {
#pragma unroll 2  // a rolled loop: sixteen columns' predicates side by side cost 180 registers and the occupancy with them
    for (int i = 0; i < 8; i++) {
      const int ca = (int)(wa & 255u), cb = (int)(wb & 255u);
      wa >>= 8, wb >>= 8;
      // branch-free: every counter adds a 0 / 1 predicate of the column (src/stats_main.cc:239-269)
      const int isup_a = (unsigned)(ca - 'A') < 26u, isup_b = (unsigned)(cb - 'A') < 26u;
      const int ua = (unsigned)(ca - 'a') < 26u ? ca - 32 : ca, ub = (unsigned)(cb - 'a') < 26u ? cb - 32 : cb;
      const int gap_a = ca == '-', gap_b = cb == '-', eq = ua == ub;
      const int both = (gap_a | gap_b) ^ 1, beq = both & eq;
      const int pur_a = (ua == 'A') | (ua == 'G'), pur_b = (ub == 'A') | (ub == 'G'), pyr_b = (ub == 'C') | (ub == 'T');
      const int same = pur_a ? pur_b : pyr_b;
      indel_a += gap_a;
      indel_b += gap_b;
      up_a += (gap_a ^ 1) & (ua != 'N') & isup_a;
      up_b += (gap_b ^ 1) & (ub != 'N') & isup_b;
      aln_b += both;
      match_b += beq;  // a != '-' && a == b: b is not '-' either
      ts += (both ^ beq) & same;
      up_m += beq & isup_a & isup_b;
      matches += beq & (ua != 'N');  // ceq (src/align.cc:29-35)
    }
  };

  // Four columns per 32-bit word, flags in bit 7 of every byte (all bytes below 0x80, so no sum carries into the next
  // byte), one v_bcnt per counter and word.  ge(v, c): byte >= c;  ne(v, c): byte != c.
  auto count_word = [&](uint32_t x, uint32_t y) {
    constexpr uint32_t O = 0x01010101u, H = 0x80808080u;
    auto ge = [](uint32_t v, uint32_t c) { return v + (0x80u - c) * O; };
    auto ne = [](uint32_t v, uint32_t c) { return (v ^ (c * O)) + 0x7Fu * O; };
    const uint32_t ux = x ^ ((ge(x, 'a') & ~ge(x, '{') & H) >> 2), uy = y ^ ((ge(y, 'a') & ~ge(y, '{') & H) >> 2);
    const uint32_t isup_x = ge(x, 'A') & ~ge(x, '['), isup_y = ge(y, 'A') & ~ge(y, '[');
    const uint32_t nd_x = ne(x, '-'), nd_y = ne(y, '-'), nN_x = ne(ux, 'N'), nN_y = ne(uy, 'N');
    const uint32_t neq = (ux ^ uy) + 0x7Fu * O;
    const uint32_t both = nd_x & nd_y, beq = both & ~neq;
    const uint32_t pur_x = ~(ne(ux, 'A') & ne(ux, 'G')), pur_y = ~(ne(uy, 'A') & ne(uy, 'G'));
    const uint32_t pyr_y = ~(ne(uy, 'C') & ne(uy, 'T'));
    const uint32_t same = (pur_x & pur_y) | (~pur_x & pyr_y);
    indel_a += __popc(~nd_x & H);
    indel_b += __popc(~nd_y & H);
    up_a += __popc(isup_x & nN_x & H);  // an upper-case letter is not '-'
    up_b += __popc(isup_y & nN_y & H);
    aln_b += __popc(both & H);
    match_b += __popc(beq & H);
    ts += __popc(both & neq & same & H);
    up_m += __popc(beq & isup_x & isup_y & H);
    matches += __popc(beq & nN_x & H);
  };
  // ('-', '-') columns count one indel on each side and nothing else: taken back per unit
  auto count = [&](uint64_t wa, uint64_t wb, int cnt) {
    indel_a -= 8 - cnt;
    indel_b -= 8 - cnt;
    if (((wa | wb) & 0x8080808080808080ULL) == 0) {
      count_word((uint32_t)wa, (uint32_t)wb);
      count_word((uint32_t)(wa >> 32), (uint32_t)(wb >> 32));
    } else {
      count_scalar(wa, wb);
    }
  };

  uint32_t w_next = lane < n_cigar ? cg[lane] : 0u;
  for (int base = 0; base < n_cigar; base += 64) {
    const int k = base + lane;
    const uint32_t w = w_next;
    w_next = k + 64 < n_cigar ? cg[k + 64] : 0u;  // the next chunk's runs travel while this chunk's columns are counted
    const int op = (int)(w & 15u);
    const int len = k < n_cigar ? (int)(w >> 4) : 0;
    // op 0 = 'M', 1 = 'D' (consumes a only), 2 = 'I' (consumes b only); anything else is refused
    const int bad_l = op > 2 || len > (a_len > b_len ? a_len : b_len);
    if (__any(bad_l)) {
      bad = 1;
      break;
    }
    const int adv_a = op != 2 ? len : 0, adv_b = op != 1 ? len : 0, nunit = (len + 7) >> 3;
    gaps += k < n_cigar && op != 0;  // zero-length runs count (src/align.cc:301-306)
    gap_bases += op != 0 ? len : 0;
    const int in_u = stats_wave_scan(nunit), in_a = stats_wave_scan(adv_a), in_b = stats_wave_scan(adv_b);
    const int total = __builtin_amdgcn_readlane(in_u, 63), tot_a = __builtin_amdgcn_readlane(in_a, 63),
              tot_b = __builtin_amdgcn_readlane(in_b, 63);
    span_l += len;  // summed over the lanes at the end
    if (ia + tot_a > a_len || ib + tot_b > b_len) {  // the reference would read past its strings
      bad = 1;
      break;
    }
    unit[lane] = in_u - nunit;
    sl[lane] = len;
    // a run that does not consume a sequence keeps no offset; the sign bit says so
    sa[lane] = op != 2 ? ia + in_a - adv_a : -1;
    sb[lane] = op != 1 ? ib + in_b - adv_b : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int u0 = 0; u0 < total; u0 += 128) {  // uniform trip count: lanes past the last unit are masked inside fetch()
      uint64_t wa0, wb0, wa1, wb1;
      int c0, c1;
      const bool two = u0 + 64 < total;  // uniform
      fetch(u0 + lane, total, wa0, wb0, c0);
      if (two) fetch(u0 + 64 + lane, total, wa1, wb1, c1);
      count(wa0, wb0, c0);
      if (two) count(wa1, wb1, c1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ia += tot_a, ib += tot_b;
  }
  const int w_[12] = {indel_a, indel_b, aln_b, match_b, ts, up_a, up_b, up_m, matches, gaps, gap_bases, span_l};
#pragma unroll
  for (int i = 0; i < 12; i++) v[i] = stats_wave_sum(w_[i]);
  return bad;
}

// ---- four short alignments per wavefront (round 4): a GROUP of sixteen lanes per alignment, sixteen runs at a time ----
// The scans, the unit search and the closing sums of one wavefront then serve four alignments: a 1 kb alignment is ~50 runs
// and ~130 units -- a wavefront of its own spends two thirds of its instructions around the counting.  Same arithmetic as
// stats_count_alignment; every wave-level step is a row-level one (a DPP row IS sixteen lanes), loop bounds are the
// maxima over the four groups, and a group that has run out of runs or units idles with ('-', '-') columns it takes back.
constexpr uint32_t STATS_GROUP_MAX = 32;  // runs of an alignment that shares its wavefront (SDF_STATS_GROUP_MAX; 0: never)

__device__ __forceinline__ int stats_row_scan(int v) {  // inclusive prefix sum inside each row of sixteen lanes
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  return v;
}
__device__ __forceinline__ int stats_row_last(int v, int lane) {  // lane 15 of the lane's own row
  return __builtin_amdgcn_ds_bpermute(((lane | 15) << 2), v);
}
__device__ __forceinline__ int stats_wave_max(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

// T: the task of this lane's group (n_cigar = 0, a_len = b_len = 0 for a group without one).  v: the group's sums.
__device__ __forceinline__ int stats_count_groups(const sdf_stats_task &T, const char *__restrict__ pool,
                                                  const uint32_t *__restrict__ cigars, int *unit, int *sa, int *sb, int *sl,
                                                  const int lane, int (&v)[12]) {
  const int gl = lane & 15, g0 = lane & ~15;
  const char *a = pool + T.a_off, *b = pool + T.b_off;
  const uint32_t *cg = cigars + T.cigar_off;
  const int n_cigar = (int)T.n_cigar, a_len = (int)T.a_len, b_len = (int)T.b_len;
  const bool wide_a = a_len >= 8, wide_b = b_len >= 8;
  int indel_a = 0, indel_b = 0, aln_b = 0, match_b = 0, ts = 0, up_a = 0, up_b = 0, up_m = 0;
  int matches = 0, gaps = 0, gap_bases = 0;
  int ia = 0, ib = 0, bad = 0;  // uniform inside a group
  int span_l = 0;

  auto count_word = [&](uint32_t x, uint32_t y) {
    constexpr uint32_t O = 0x01010101u, H = 0x80808080u;
    auto ge = [](uint32_t vv, uint32_t c) { return vv + (0x80u - c) * O; };
    auto ne = [](uint32_t vv, uint32_t c) { return (vv ^ (c * O)) + 0x7Fu * O; };
    const uint32_t ux = x ^ ((ge(x, 'a') & ~ge(x, '{') & H) >> 2), uy = y ^ ((ge(y, 'a') & ~ge(y, '{') & H) >> 2);
    const uint32_t isup_x = ge(x, 'A') & ~ge(x, '['), isup_y = ge(y, 'A') & ~ge(y, '[');
    const uint32_t nd_x = ne(x, '-'), nd_y = ne(y, '-'), nN_x = ne(ux, 'N'), nN_y = ne(uy, 'N');
    const uint32_t neq = (ux ^ uy) + 0x7Fu * O;
    const uint32_t both = nd_x & nd_y, beq = both & ~neq;
    const uint32_t pur_x = ~(ne(ux, 'A') & ne(ux, 'G')), pur_y = ~(ne(uy, 'A') & ne(uy, 'G'));
    const uint32_t pyr_y = ~(ne(uy, 'C') & ne(uy, 'T'));
    const uint32_t same = (pur_x & pur_y) | (~pur_x & pyr_y);
    indel_a += __popc(~nd_x & H);
    indel_b += __popc(~nd_y & H);
    up_a += __popc(isup_x & nN_x & H);
    up_b += __popc(isup_y & nN_y & H);
    aln_b += __popc(both & H);
    match_b += __popc(beq & H);
    ts += __popc(both & neq & same & H);
    up_m += __popc(beq & isup_x & isup_y & H);
    matches += __popc(beq & nN_x & H);
  };
  auto count_scalar = [&](uint64_t wa, uint64_t wb) {  // (the statement of src/stats_main.cc:239-269, for bytes outside ASCII)
#pragma unroll 2
    for (int i = 0; i < 8; i++) {
      const int ca = (int)(wa & 255u), cb = (int)(wb & 255u);
      wa >>= 8, wb >>= 8;
      const int isup_a = (unsigned)(ca - 'A') < 26u, isup_b = (unsigned)(cb - 'A') < 26u;
      const int ua = (unsigned)(ca - 'a') < 26u ? ca - 32 : ca, ub = (unsigned)(cb - 'a') < 26u ? cb - 32 : cb;
      const int gap_a = ca == '-', gap_b = cb == '-', eq = ua == ub;
      const int both = (gap_a | gap_b) ^ 1, beq = both & eq;
      const int pur_a = (ua == 'A') | (ua == 'G'), pur_b = (ub == 'A') | (ub == 'G'), pyr_b = (ub == 'C') | (ub == 'T');
      const int same = pur_a ? pur_b : pyr_b;
      indel_a += gap_a;
      indel_b += gap_b;
      up_a += (gap_a ^ 1) & (ua != 'N') & isup_a;
      up_b += (gap_b ^ 1) & (ub != 'N') & isup_b;
      aln_b += both;
      match_b += beq;
      ts += (both ^ beq) & same;
      up_m += beq & isup_a & isup_b;
      matches += beq & (ua != 'N');
    }
  };
  auto count = [&](uint64_t wa, uint64_t wb, int cnt) {
    indel_a -= 8 - cnt;
    indel_b -= 8 - cnt;
    if (((wa | wb) & 0x8080808080808080ULL) == 0) {
      count_word((uint32_t)wa, (uint32_t)wb);
      count_word((uint32_t)(wa >> 32), (uint32_t)(wb >> 32));
    } else {
      count_scalar(wa, wb);
    }
  };

  const int n_max = stats_wave_max(n_cigar);
  for (int base = 0; base < n_max; base += 16) {
    const int k = base + gl;
    const bool have = k < n_cigar && !bad;
    const uint32_t w = have ? cg[k] : 0u;
    const int op = (int)(w & 15u);
    int len = have ? (int)(w >> 4) : 0;
    const int bad_l = have && (op > 2 || len > (a_len > b_len ? a_len : b_len));
    if (((__ballot(bad_l) >> g0) & 0xffffull) != 0) bad = 1, len = 0;
    const int adv_a = op != 2 ? len : 0, adv_b = op != 1 ? len : 0, nunit = (len + 7) >> 3;
    gaps += have && !bad && op != 0;  // zero-length runs count (src/align.cc:301-306)
    gap_bases += op != 0 ? len : 0;
    const int in_u = stats_row_scan(nunit), in_a = stats_row_scan(adv_a), in_b = stats_row_scan(adv_b);
    int total = stats_row_last(in_u, lane);
    const int tot_a = stats_row_last(in_a, lane), tot_b = stats_row_last(in_b, lane);
    span_l += len;
    if (ia + tot_a > a_len || ib + tot_b > b_len) bad = 1, total = 0;  // the reference would read past its strings
    unit[lane] = in_u - nunit;
    sl[lane] = len;
    sa[lane] = op != 2 ? ia + in_a - adv_a : -1;
    sb[lane] = op != 1 ? ib + in_b - adv_b : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int total_max = stats_wave_max(total);
    for (int u0 = 0; u0 < total_max; u0 += 16) {
      const int u = u0 + gl;
      const bool valid = u < total;
      int j = g0;
#pragma unroll
      for (int step = 8; step; step >>= 1)
        if (valid && unit[j + step] <= u) j += step;  // last run of the group that starts at or before unit u
      uint64_t wa = STATS_DASHES, wb = STATS_DASHES;
      int cnt = 0;
      if (valid) {
        const int d = 8 * (u - unit[j]), pa = sa[j], pb = sb[j];
        const int left = sl[j] - d;
        cnt = left < 8 ? left : 8;
        if (pa >= 0) wa = stats_fetch8(a, pa + d, a_len, wide_a);
        if (pb >= 0) wb = stats_fetch8(b, pb + d, b_len, wide_b);
        const uint64_t keep = cnt >= 8 ? ~0ULL : (1ULL << (8 * cnt)) - 1ULL;
        wa = (wa & keep) | (STATS_DASHES & ~keep);
        wb = (wb & keep) | (STATS_DASHES & ~keep);
      }
      count(wa, wb, cnt);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ia += tot_a, ib += tot_b;
  }
  const int w_[12] = {indel_a, indel_b, aln_b, match_b, ts, up_a, up_b, up_m, matches, gaps, gap_bases, span_l};
#pragma unroll
  for (int i = 0; i < 12; i++) v[i] = stats_row_last(stats_row_scan(w_[i]), lane);
  return bad;
}

__device__ __forceinline__ sdf_stats_cols stats_record(const int (&v)[12], const int bad) {
  sdf_stats_cols R;
  R.indel_a = v[0], R.indel_b = v[1], R.aln_b = v[2], R.match_b = v[3], R.mismatch_b = v[2] - v[3];
  R.transitions_b = v[4], R.transversions_b = v[2] - v[3] - v[4], R.uppercase_a = v[5], R.uppercase_b = v[6];
  R.uppercase_matches = v[7], R.matches = v[8], R.mismatches = v[2] - v[8], R.gaps = v[9], R.gap_bases = v[10];
  R.span = v[11], R.flags = bad;
  return R;
}

// One wavefront per alignment.  items / counter / cap: the list for the segments of long alignments (items == nullptr:
// every alignment is counted by its own wavefront).
// one alignment by a whole wavefront (the long ones: into segments for the second launch)
__device__ __forceinline__ void stats_one_task(const int task, const sdf_stats_task *__restrict__ tasks, const char *__restrict__ pool,
                                               const uint32_t *__restrict__ cigars, sdf_stats_cols *__restrict__ out,
                                               StatsItem *__restrict__ items, unsigned *__restrict__ counter, unsigned cap,
                                               int *s_unit_w, int *s_a_w, int *s_b_w, int *s_len_w, const int lane) {
  const sdf_stats_task T = tasks[task];
  if (items && T.n_cigar > STATS_LONG) {
    const unsigned nseg = (T.n_cigar + STATS_SEG - 1) / STATS_SEG;
    unsigned first = 0;
    if (lane == 0) first = atomicAdd(counter, nseg);
    first = (unsigned)__builtin_amdgcn_readfirstlane((int)first);
    if (first < cap && nseg <= cap - first) {
      // the segments' advances: eight runs per lane and segment; the sequences' offsets are running sums over the segments
      const uint32_t *cg = cigars + T.cigar_off;
      uint64_t ia = 0, ib = 0;
      int bad = 0;
      for (unsigned sg = 0; sg < nseg; ++sg) {
        const unsigned k0 = sg * STATS_SEG, nk = T.n_cigar - k0 < STATS_SEG ? T.n_cigar - k0 : STATS_SEG;
        // (512 runs of up to 2^24 columns add up to 2^33: the low twelve bits and the rest of every length are summed apart --
        // each sum stays below 2^21 -- and put together in 64 bits; a segment of 2^31 columns or more does not fit its item)
        int da_lo = 0, da_hi = 0, db_lo = 0, db_hi = 0, wrong = 0;
#pragma unroll
        for (unsigned q = 0; q < STATS_SEG / 64; ++q) {
          const unsigned k = q * 64 + lane;
          const uint32_t w = k < nk ? cg[k0 + k] : 0u;
          const uint32_t op = w & 15u, len = w >> 4;
          wrong |= op > 2 || len > (1u << 24);
          da_lo += op != 2 ? (int)(len & 4095u) : 0, da_hi += op != 2 ? (int)(len >> 12) : 0;
          db_lo += op != 1 ? (int)(len & 4095u) : 0, db_hi += op != 1 ? (int)(len >> 12) : 0;
        }
        const int64_t da64 = ((int64_t)stats_wave_sum(da_hi) << 12) + stats_wave_sum(da_lo);
        const int64_t db64 = ((int64_t)stats_wave_sum(db_hi) << 12) + stats_wave_sum(db_lo);
        const int da = (int)(da64 > 0x7fffffff ? 0x7fffffff : da64), db = (int)(db64 > 0x7fffffff ? 0x7fffffff : db64);
        bad |= __any(wrong) || da64 > 0x7ffffffe || db64 > 0x7ffffffe || ia + (uint64_t)da64 > T.a_len || ib + (uint64_t)db64 > T.b_len;
        if (lane == 0) {
          StatsItem it;
          it.t.a_off = T.a_off + ia, it.t.b_off = T.b_off + ib;
          it.t.a_len = (uint32_t)da, it.t.b_len = (uint32_t)db;
          it.t.cigar_off = T.cigar_off + k0, it.t.n_cigar = nk, it.t.reserved = 0;
          it.task = bad ? 0xffffffffu : (uint32_t)task;  // (nothing after a segment that does not fit is counted)
          it.pad = 0;
          items[first + sg] = it;
        }
        ia += (uint64_t)(bad ? 0 : da), ib += (uint64_t)(bad ? 0 : db);
      }
      if (lane == 0) {
        const int zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        out[task] = stats_record(zero, bad);  // the segments add to it
      }
      return;
    }
    // the list is full: its slots (if any) stay empty and this wavefront counts the alignment itself
    for (unsigned sg = lane; first < cap && sg < cap - first; sg += 64) items[first + sg].task = 0xffffffffu;
  }
  int v[12];
  const int bad = stats_count_alignment(T, pool, cigars, s_unit_w, s_a_w, s_b_w, s_len_w, lane, v);
  if (lane == 0) out[task] = stats_record(v, bad);
}

// One wavefront per alignment -- except that four consecutive SHORT alignments (at most `group_max` runs each: a chunk or two
// of sixteen) are taken side by side by one wavefront, in its four rows of sixteen lanes.  items / counter / cap: the list
// for the segments of long alignments (items == nullptr: every alignment is counted whole).
__global__ __launch_bounds__(64 * STATS_WAVES, 6) void stats_columns_kernel(const sdf_stats_task *__restrict__ tasks, int n,
                                                                         const char *__restrict__ pool,
                                                                         const uint32_t *__restrict__ cigars,
                                                                         sdf_stats_cols *__restrict__ out,
                                                                         StatsItem *__restrict__ items,
                                                                         unsigned *__restrict__ counter, unsigned cap,
                                                                         unsigned group_max) {
  __shared__ int s_unit[STATS_WAVES][64], s_a[STATS_WAVES][64], s_b[STATS_WAVES][64], s_len[STATS_WAVES][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = blockIdx.x * STATS_WAVES + wv;  // (four consecutive tasks = the four wavefronts of a workgroup)
  if (task >= n) return;  // whole wavefronts leave; the kernel has no workgroup barrier
  // every wavefront of the quad looks at all four alignments: all short -> the first wavefront takes them side by side and
  // the other three leave; else every wavefront takes its own
  const int task0 = task & ~3, mine = task0 + (lane >> 4);
  sdf_stats_task T;
  if (mine < n) {
    T = tasks[mine];
  } else {
    T.a_off = T.b_off = T.cigar_off = 0;
    T.a_len = T.b_len = T.n_cigar = T.reserved = 0;
  }
  if (group_max && !__any(T.n_cigar > group_max)) {
    if (task != task0) return;
    int v[12];
    const int bad = stats_count_groups(T, pool, cigars, s_unit[wv], s_a[wv], s_b[wv], s_len[wv], lane, v);
    if ((lane & 15) == 0 && mine < n) out[mine] = stats_record(v, bad);
    return;
  }
  stats_one_task(task, tasks, pool, cigars, out, items, counter, cap, s_unit[wv], s_a[wv], s_b[wv], s_len[wv], lane);
}

// The segments of the long alignments: wavefront g of the grid takes items g, g + G, ... and adds their counters up.
__global__ __launch_bounds__(64 * STATS_WAVES, 6) void stats_segments_kernel(const StatsItem *__restrict__ items,
                                                                          const unsigned *__restrict__ counter, unsigned cap,
                                                                          const char *__restrict__ pool,
                                                                          const uint32_t *__restrict__ cigars,
                                                                          sdf_stats_cols *__restrict__ out) {
  __shared__ int s_unit[STATS_WAVES][64], s_a[STATS_WAVES][64], s_b[STATS_WAVES][64], s_len[STATS_WAVES][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned total = *counter < cap ? *counter : cap;
  for (unsigned i = blockIdx.x * STATS_WAVES + wv; i < total; i += gridDim.x * STATS_WAVES) {
    const StatsItem it = items[i];
    if (it.task == 0xffffffffu) continue;  // (wave-uniform)
    int v[12];
    const int bad = stats_count_alignment(it.t, pool, cigars, s_unit[wv], s_a[wv], s_b[wv], s_len[wv], lane, v);
    const sdf_stats_cols R = stats_record(v, bad);
    const int32_t *src = reinterpret_cast<const int32_t *>(&R);
    int32_t *dst = reinterpret_cast<int32_t *>(&out[it.task]);
    if (lane < 15) atomicAdd(dst + lane, src[lane]);
    if (lane == 15 && bad) atomicOr(dst + 15, 1);
  }
}

}  // namespace sdf
