// Resident sequences: DP tasks that point INTO characters a batch of candidate pairs already has in HBM.
//
// The reference reads the bases of every DP call in place (src/align.cc:49-57: `align_helper` takes two strings the
// caller cut out of the pair's sequences and maps through align_dna, src/align.cc:80-84).  The stage driver uploads a
// super-batch's FASTA characters once, for the seed anchors (anchors.hip); its DP rounds then name their tasks as ranges of
// that pool and this kernel does what `align_dna` + sdf_pack_codes do on the host: characters -> codes (ACGT, either case,
// 0..3; anything else the wildcard 4, src/common.h:60-70,91) -> the packed per-task layout every DP kernel reads
// (include/sedef_hip.h: ceil(len/16) words of 2-bit codes, then ceil(len/32) words of N mask).
#include <hip/hip_runtime.h>

#include "sdf_internal.h"

namespace sdf {

struct PackRec {     // one DP task's two character ranges and where its packed words go (32 bytes)
  int64_t q_byte;    // first character of the query range in the pool
  int64_t t_byte;
  int64_t q_word;    // first packed word of the query; the target's words follow the query's
  int32_t qlen, tlen;
};

// Sixteen lanes per sequence (a task is two sequences), a lane per group of 32 bases -- two code words and a mask word.
// SEDEF's tasks are short (708,600 of ~25 bases in a round of the chr1-sized run) with a few of up to 60,000 bases
// (Align::MAX_KSW_SEQ_LEN): the lanes of a group stride over a long sequence.
__global__ void __launch_bounds__(256) pack_chars_kernel(const PackRec *__restrict__ recs, long long n_seq,
                                                         const char *__restrict__ pool, uint32_t *__restrict__ out) {
  const int sub = threadIdx.x & 15;
  const long long seq = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (seq >= n_seq) return;
  const PackRec r = recs[seq >> 1];
  const bool is_t = (seq & 1) != 0;
  const int len = is_t ? r.tlen : r.qlen;
  const char *src = pool + (is_t ? r.t_byte : r.q_byte);
  const int q_words = ((r.qlen + 15) >> 4) + ((r.qlen + 31) >> 5);
  uint32_t *dst = out + r.q_word + (is_t ? q_words : 0);
  const int ncode = (len + 15) >> 4;
  for (int g = sub; 32 * g < len; g += 16) {
    uint32_t c0 = 0, c1 = 0, m = 0;
    const int lim = min(32, len - 32 * g);
    for (int b = 0; b < lim; ++b) {
      const unsigned c = (unsigned char)src[32 * g + b] & 0x5fu;  // (& 127 as the reference's table index, then upper case)
      const unsigned code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
      if (code == 4u) m |= 1u << b;
      else if (b < 16) c0 |= code << (2 * b);
      else c1 |= code << (2 * (b - 16));
    }
    dst[2 * g] = c0;
    if (32 * g + 16 < len) dst[2 * g + 1] = c1;
    dst[ncode + g] = m;
  }
}

}  // namespace sdf
