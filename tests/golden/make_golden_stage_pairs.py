#!/usr/bin/env python3
"""Generates tests/golden/stage_pairs_kat.json.gz: what the REFERENCE produces for candidate pairs and whole bucket files.

Build container only.  src/chain.cc, src/refine.cc and src/align_main.cc cannot be compiled here (Boost.ICL through
src/search.h:22-23), so the control flow of fast_align / chain_anchors' sweep / refine_chains / the driver's ordering and
remap is restated from those files in tests/bruteforce.py (StageModel, definition level: not derived from this
repository's host code), and EVERY object on the way is the reference's own, compiled unmodified
(oracle/_ref/libref_align.so): Alignment(anchors) of each chain, Alignment::merge, the guide alignment with side extension,
update_from_alignment, SegmentTree activate / deactivate / rmq, FastaReference::get_sequence, Hit::to_bed.  Seed anchors
come from the brute-force definition (tests/bruteforce.py: anchors_bruteforce).

Sections (inputs and expected outputs only):
  pairs   -- fast_align(query, ref, orig, k) per candidate pair: final hits (coordinates, CIGAR, counters) + the first
             round's chain hits; kinds cover forward / rc, same-chromosome pairs incl. overlapping windows
             (src/refine.cc:42-53,80-88), tandem copies, far gaps with equal and unequal sides (src/align.cc:131-139), gaps
             beyond Refine::MAX_GAP, chains on either side of the 489.99999999999994 threshold (src/chain.cc:233-238),
             score ties of the O(n^2) pass (src/refine.cc:93), soft-masked and N-rich sequences, long pairs.
  chains  -- chain_anchors(anchors) -> (path, boundaries) of those pairs and of tie-rich synthetic anchor sets, from the
             sweep run on the reference's own SegmentTree.
  stages  -- FASTA + .fai + bucket file -> the exact stdout of `sedef align generate`.
"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.binding import ReferenceAlign  # noqa: E402
import bruteforce  # noqa: E402
import hostgen  # noqa: E402


class Orig:
    def __init__(self, c):
        self.qname, self.rname, self.q_rc, self.r_rc, self.qs, self.rs = (c["qname"], c["rname"], c["q_rc"], c["r_rc"],
                                                                           c["qstart"], c["rstart"])


def hit_record(ref, h):
    g = ref.tab_get(h)
    return [g["qs"], g["qe"], g["rs"], g["re"], ref.tab_cigar(h), g["matches"], g["mismatches"], g["gaps"], g["gap_bases"]]


def tie_rich_anchors(rng):
    """Anchors on a coarse grid: many equal stored scores in the tree's query window (src/chain.cc:157-176)."""
    n = int(rng.integers(2, 60))
    step = int(rng.choice([12, 15, 20, 40]))
    seen, out = set(), []
    same = rng.random() < 0.7  # one length and one case for all: chain heads of equal score on equal anti-diagonals
    l0, hu0 = int(rng.choice([11, 12, step - 1, step])), int(rng.integers(0, 2))
    for _ in range(n):
        q, r = int(rng.integers(0, 30)) * step, int(rng.integers(0, 30)) * step
        l = l0 if same else int(rng.choice([11, 12, step - 1, step]))
        if (q, r) in seen:
            continue
        seen.add((q, r))
        out.append((q, r, l, hu0 if same else int(rng.integers(0, 2))))
    out.sort(key=lambda a: (a[0], a[1]))
    return out


def synthetic_chains(ref):
    rng = np.random.default_rng(77001)
    m = bruteforce.StageModel(ref)
    out = []
    for _ in range(150):
        a = tie_rich_anchors(rng)
        if len(a) < 2:
            continue
        path, bounds = m.chain_anchors(a)
        cc = bruteforce.ChainCheck(a)
        cc.check(path, bounds)
        out.append(dict(anchors=[list(x) for x in a], path=list(path), bounds=[list(b) for b in bounds], ties=bool(cc.ties)))
    return out


def write(out, obj):
    blob = json.dumps(obj, separators=(",", ":")).encode()
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(blob)


def main():
    ref = ReferenceAlign()
    t0 = time.time()
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage_pairs_kat.json.gz")
    if "--synthetic-chains-only" in sys.argv:  # (the pairs and stages sections take minutes: keep them, redo the synthetic chains)
        with gzip.open(out_path, "rb") as f:
            old = json.loads(f.read().decode())
        old["chains"] = [c for c in old["chains"] if "ties" not in c] + synthetic_chains(ref)
        write(out_path, old)
        print("rewrote the synthetic chain cases: %d chain cases, %d with ties" % (
            len(old["chains"]), sum(bool(c.get("ties")) for c in old["chains"])))
        return
    rng = np.random.default_rng(20261004)
    per_kind = dict(plain=22, indel=18, rc=18, same_chr=16, self_overlap=22, tandem=22, far_equal=14, far_unequal=16,
                    far_cut=10, threshold=30, dp_tie=16, low_upper=12, n_runs=12, long=6)
    pairs, chains, notes_total, ambiguous = [], [], {}, 0
    for kind in hostgen.STAGE_PAIR_KINDS:
        for _ in range(per_kind[kind]):
            c = hostgen.stage_pair_case(rng, kind)
            m = bruteforce.StageModel(ref)
            try:
                ids = m.fast_align(c["query"], c["ref"], Orig(c), 11)
            except bruteforce.Ambiguous:
                ambiguous += 1
                continue
            c["kmer"] = 11
            c["first_round"] = [hit_record(ref, h)[:4] + [ref.tab_get(h)["jaccard"]] for h in m.last_first_round]
            c["expect"] = [hit_record(ref, h) for h in ids]
            c["notes"] = dict(m.notes)
            pairs.append(c)
            for k, v in m.notes.items():
                key = k if not k.startswith("threshold_") else k[:14]
                notes_total[key] = notes_total.get(key, 0) + v
            if 2 <= len(m.last_anchors) <= 400 and len(chains) < 120:
                path, bounds = m.last_chain
                chains.append(dict(anchors=[list(a) for a in m.last_anchors], path=list(path), bounds=[list(b) for b in bounds]))
    chains += synthetic_chains(ref)
    stages = []
    for seed, kw in ((11, {}), (12, dict(chrom_lens=(70_000, 40_000), line_blens=(61, 17), n_pairs=14, max_len=5000)),
                     (13, dict(chrom_lens=(150_000,), line_blens=(80,), n_pairs=12, max_len=22000))):
        fx = hostgen.make_stage_fixture(seed, **kw)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "g.fa")  # no .fai next to it (oracle/ref_align_driver.cc: ref_fasta_get)
            open(path, "w").write(fx["fasta"])
            m = bruteforce.StageModel(ref)
            out = m.generate(path, fx["fai"], fx["bed"], 11)
        fx["kmer"] = 11
        fx["expect"] = out
        fx["notes"] = dict(m.notes)
        stages.append(fx)
        print("stage seed %d: %d seeds -> %d lines  %s" % (seed, fx["bed"].count("\n"), len(out), m.notes), flush=True)
    out = out_path
    write(out, dict(source="tests/bruteforce.py StageModel over the reference's Alignment / Hit / SegmentTree / "
                           "FastaReference classes (oracle/_ref/libref_align.so)",
                    pairs=pairs, chains=chains, stages=stages, notes=notes_total))
    print("wrote %s: %d pairs (%d ambiguous skipped), %d chain cases, %d stages, %d bytes, %.0f s" % (
        out, len(pairs), ambiguous, len(chains), len(stages), os.path.getsize(out), time.time() - t0))
    print("notes:", json.dumps(notes_total, sort_keys=True))
    print("pairs with hits: %d, without: %d" % (sum(bool(p["expect"]) for p in pairs), sum(not p["expect"] for p in pairs)))


if __name__ == "__main__":
    main()
