"""Generates tests/golden/stats_columns_kat.json.gz from the REFERENCE's own Alignment(fa, fb, cigar) constructor
(src/align.cc:90-106 -> populate_nice_alignment :274-315), compiled from /root/reference by oracle/Makefile `refalign`.
Run in the build container only (the reference does not travel):  python tests/golden/make_golden_stats_columns.py

Each case: a, b (FASTA characters, mixed case, N runs), cigar string, and what the reference returned: the column strings
align_a / align_b and [matches, mismatches, gaps, gap_bases, span]."""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.binding import ReferenceAlign  # noqa: E402
from util import random_stats_case  # noqa: E402


def main():
    ref = ReferenceAlign()
    rng = np.random.default_rng(20261003)
    cases = []
    for k in range(400):
        a, b, runs = random_stats_case(rng, k)
        cig = "".join("%d%s" % (l, "MDI"[op]) for op, l in runs)
        aa, ab, cnt = ref.alignment_from_cigar(a, b, cig)
        cases.append({"a": a, "b": b, "cigar": cig, "align_a": aa, "align_b": ab, "counts": cnt})
    out = os.path.join(ROOT, "tests", "golden", "stats_columns_kat.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(cases, separators=(",", ":")).encode())
    print(out, os.path.getsize(out), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
