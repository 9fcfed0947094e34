/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-alignment columns of `sedef stats generate` (scope row f4).
 *
 * Plain-C restatement of two pieces of the reference, written the way the reference computes them (the CIGAR is
 * first expanded into the column strings align_a / align_b, then the columns are walked):
 *   - populate_nice_alignment: expansion + the AlignmentError counters        (reference: src/align.cc:274-315, ceq :29-35)
 *   - process(): the column loop behind output columns 15-21 and 27-29         (reference: src/stats_main.cc:228-270)
 *
 * Parity status: the expansion and the four AlignmentError counters are PINNED -- tests/test_stats_columns.py checks
 * them against the reference's own Alignment(fa, fb, cigar) constructor (oracle/_ref/libref_align.so, built from
 * src/align.cc) live when /root/reference is present and against the committed golden vectors it produced
 * (tests/golden/stats_columns_kat.json.gz).  The column loop of process() is PARITY UNPINNED: src/stats_main.cc includes
 * boost/dynamic_bitset.hpp, Boost is absent from this image, so that translation unit cannot be compiled here; the
 * loop below restates its 40 lines over column strings that are proven identical to the reference's.
 *
 * Only tests/ may call this code.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int ceq(char a, char b) { /* src/align.cc:29-35 */
  if (a == '-' || b == '-') return 0;
  if (toupper((unsigned char)a) == 'N' || toupper((unsigned char)b) == 'N') return 0;
  return toupper((unsigned char)a) == toupper((unsigned char)b);
}

/* cigar: runs len << 4 | op, op 0 = 'M', 1 = 'D', 2 = 'I' (the reference's letters, src/align.cc:59-63).
 * out[16]: indel_a indel_b alnB matchB mismatchB transitionsB transversionsB uppercaseA uppercaseB uppercaseMatches
 *          matches mismatches gaps gap_bases span flags.
 * align_a / align_b (optional, capacity alen + blen + 1 each) receive the column strings.
 * Returns 0, or -1 when the CIGAR consumes more than the sequences hold (flags = 1; the reference asserts / reads
 * past its strings there). */
int sdfo_stats_columns(const char *a, int alen, const char *b, int blen, const uint32_t *cigar, int n_cigar,
                       int32_t *out, char *align_a, char *align_b) {
  memset(out, 0, 16 * sizeof(int32_t));
  long need_a = 0, need_b = 0;
  for (int k = 0; k < n_cigar; k++) {
    const int op = cigar[k] & 15, len = cigar[k] >> 4;
    if (op > 2) { out[15] = 1; return -1; }
    if (op != 2) need_a += len;
    if (op != 1) need_b += len;
  }
  if (need_a > alen || need_b > blen) { out[15] = 1; return -1; }
  char *ca = align_a ? align_a : (char *)malloc((size_t)alen + blen + 1);
  char *cb = align_b ? align_b : (char *)malloc((size_t)alen + blen + 1);
  /* src/align.cc:278-298 */
  int ia = 0, ib = 0, n = 0;
  for (int k = 0; k < n_cigar; k++) {
    const char op = "MDI"[cigar[k] & 15];
    const int len = cigar[k] >> 4;
    for (int i = 0; i < len; i++) {
      cb[n] = op != 'D' ? b[ib++] : '-';
      ca[n] = op != 'I' ? a[ia++] : '-';
      n++;
    }
  }
  ca[n] = cb[n] = 0;
  /* src/align.cc:300-314 */
  int gaps = 0, gap_bases = 0, matches = 0, mismatches = 0;
  for (int k = 0; k < n_cigar; k++)
    if ((cigar[k] & 15) != 0) {
      gaps++;
      gap_bases += cigar[k] >> 4;
    }
  for (int i = 0; i < n; i++)
    if (ca[i] != '-' && cb[i] != '-') {
      if (ceq(ca[i], cb[i])) matches++;
      else mismatches++;
    }
  /* src/stats_main.cc:228-270 */
  int indel_a = 0, indel_b = 0, alignB = 0, matchB = 0, mismatchB = 0, transitionsB = 0, transversionsB = 0;
  int uppercaseA = 0, uppercaseB = 0, uppercaseMatches = 0;
  for (int i = 0; i < n; i++) {
    const char x = (char)toupper((unsigned char)ca[i]), y = (char)toupper((unsigned char)cb[i]);
    indel_a += x == '-';
    indel_b += y == '-';
    matchB += x != '-' && x == y;
    uppercaseA += ca[i] != '-' && toupper((unsigned char)ca[i]) != 'N' && isupper((unsigned char)ca[i]);
    uppercaseB += cb[i] != '-' && toupper((unsigned char)cb[i]) != 'N' && isupper((unsigned char)cb[i]);
    if (x != '-' && y != '-') {
      alignB += 1;
      if (x != y) {
        mismatchB += 1;
        if (x == 'A' || x == 'G') {
          transitionsB += y == 'A' || y == 'G';
          transversionsB += !(y == 'A' || y == 'G');
        } else {
          transitionsB += y == 'C' || y == 'T';
          transversionsB += !(y == 'C' || y == 'T');
        }
      } else if (isupper((unsigned char)ca[i]) && isupper((unsigned char)cb[i])) {
        uppercaseMatches++;
      }
    }
  }
  const int32_t v[16] = {indel_a, indel_b, alignB, matchB, mismatchB, transitionsB, transversionsB, uppercaseA,
                         uppercaseB, uppercaseMatches, matches, mismatches, gaps, gap_bases, n, 0};
  memcpy(out, v, sizeof(v));
  if (!align_a) free(ca);
  if (!align_b) free(cb);
  return 0;
}
