#!/usr/bin/env python3
"""Generates tests/golden/host_align_kat.json.gz from the REFERENCE's Alignment / Hit classes.

Build container only: compiles src/align.cc, src/hit.cc, ... unmodified (oracle/Makefile `refalign`) and
records, for seeded inputs, what the reference returns for (1) Alignment(fa, fb) -- CIGAR string and the
populate_nice_alignment counters -- and (2) chain alignments + merge + guide alignment with side
extension + Hit::to_bed, (3) merge() of src/merge.cc on random seed hits.  Inputs and expected outputs only."""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.binding import ReferenceAlign  # noqa: E402
import hostgen  # noqa: E402
from sedef_amd import host  # noqa: E402


def main():
    ref = ReferenceAlign()
    rng = np.random.default_rng(777)
    pairs, guides = [], []
    for it in range(60):
        a = hostgen.rseq(rng, int(rng.integers(1, 500)), 0.01 if it % 3 == 0 else 0)
        b = hostgen.mut(rng, a, rng.random() * 0.2)
        cig, cnt = ref.alignment_pair(a, b)
        pairs.append(dict(a=a, b=b, cigar=cig, counts=cnt))
    while len(guides) < 30:
        c = hostgen.chain_case(rng, host)
        if c is None:
            continue
        q, r, spec, side = c
        guides.append(dict(q=q, r=r, spec=spec, side=side, expect=ref.guide_from_chains(q, r, spec, side)))
    merges = []
    for _ in range(40):
        lines, spec = hostgen.merge_case(rng)
        merges.append(dict(lines=lines, expect=ref.merge(spec, 250)))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_align_kat.json.gz")
    blob = json.dumps(dict(source="reference src/align.cc + src/hit.cc via oracle/_ref/libref_align.so",
                           pairs=pairs, guides=guides, merges=merges), separators=(",", ":")).encode()
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(blob)
    print("wrote %s: %d pairs, %d guides, %d bytes" % (out, len(pairs), len(guides), os.path.getsize(out)))


if __name__ == "__main__":
    main()
