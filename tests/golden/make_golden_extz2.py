#!/usr/bin/env python3
"""Generates tests/golden/extz2_kat.json.gz from the REFERENCE kernel.

Run in the build container only (needs /root/reference): it compiles the reference's
extern/ksw2_extz2_sse.cc unmodified (oracle/Makefile `ref` target), calls ksw_extz2_sse
(extern/ksw2.h:50) on seeded inputs and records every ksw_extz_t field plus the CIGAR.
The fixture is data only: inputs and the reference's outputs.
"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference, cigar_to_str, mutate, random_codes, sedef_mat  # noqa: E402

ALPHA = "ACGTN"


def to_s(codes):
    return "".join(ALPHA[c] for c in np.asarray(codes).tolist())


def main():
    ref = Reference()
    rng = np.random.default_rng(20261002)
    cases = []

    def add(tag, q, t, w=-1, zdrop=-1, flag=0, match=5, mismatch=-4, gapo=40, gape=1):
        r = ref.extz2(q, t, mat=sedef_mat(match, mismatch), gapo=gapo, gape=gape, w=w, zdrop=zdrop,
                      flag=flag)
        cases.append(dict(tag=tag, q=to_s(q), t=to_s(t), w=w, zdrop=zdrop, flag=flag, match=match,
                          mismatch=mismatch, gapo=gapo, gape=gape,
                          expect=dict(max=r["max"], zdropped=r["zdropped"], max_q=r["max_q"],
                                      max_t=r["max_t"], mqe=r["mqe"], mqe_t=r["mqe_t"],
                                      mte=r["mte"], mte_q=r["mte_q"], score=r["score"],
                                      cigar=cigar_to_str(r["cigar"]))))

    lens = [1, 2, 15, 16, 17, 31, 33, 209, 210, 500, 1000, 1001]
    bands = [-1, 0, 1, 15, 16, 17, 64, 128, 512]
    # (1) related pairs over the length x band grid, 10 % divergence
    for ql in lens:
        for w in bands:
            q = random_codes(rng, ql)
            add("grid", q, mutate(rng, q), w=w)
    # (2) unrelated lengths
    for _ in range(40):
        ql, tl = int(rng.choice(lens)), int(rng.choice(lens))
        add("unrelated", random_codes(rng, ql), random_codes(rng, tl), w=int(rng.choice(bands)))
    # (3) divergence sweep 10-25 %, N content
    for div in (0.10, 0.15, 0.20, 0.25):
        for ql in (210, 500, 1000):
            q = random_codes(rng, ql, n_frac=0.02)
            t = mutate(rng, q, sub=div * 0.6, dele=div * 0.2, ins=div * 0.2)
            for w in (-1, 64, 128):
                add("div%.2f" % div, q, t, w=w)
    # (4) all-N, N runs, homopolymers
    for ql in (1, 16, 33, 210):
        add("allN", np.full(ql, 4, np.uint8), np.full(ql + 3, 4, np.uint8))
        add("allN_vs_seq", np.full(ql, 4, np.uint8), random_codes(rng, ql), w=16)
        add("homopolymer", np.zeros(ql, np.uint8), np.zeros(ql + 5, np.uint8), w=17)
        add("homo_vs_other", np.zeros(ql, np.uint8), np.full(ql, 3, np.uint8))
    for _ in range(8):
        q = random_codes(rng, 500)
        t = mutate(rng, q)
        a = int(rng.integers(0, 400))
        q[a:a + 60] = 4
        b = int(rng.integers(0, len(t) - 80))
        t[b:b + 40] = 4
        add("Nrun", q, t, w=int(rng.choice([-1, 64, 128])))
    # (5) big indel near / beyond the band edge
    for w in (16, 64, 128):
        for gap in (w // 2 - 1, w // 2, w // 2 + 1, w - 1, w, w + 1, 2 * w):
            q = random_codes(rng, 600)
            t = mutate(rng, q, sub=0.03, dele=0.0, ins=0.0)
            k = int(rng.integers(100, 400))
            if rng.random() < 0.5:
                t = np.concatenate([t[:k], random_codes(rng, gap), t[k:]])
            else:
                t = np.concatenate([t[:k], t[k + gap:]])
            add("indel_edge", q, t, w=w)
    # (6) flags and z-drop
    for flag in (0x01, 0x02, 0x04, 0x08, 0x18, 0x40, 0x80, 0x42, 0xC0):
        for w in (-1, 33):
            q = random_codes(rng, 300)
            add("flag", q, mutate(rng, q), w=w, flag=flag, zdrop=int(rng.choice([-1, 100])))
    for zd in (0, 10, 50, 200):
        q = random_codes(rng, 400)
        t = np.concatenate([mutate(rng, q[:200]), random_codes(rng, 250)])
        add("zdrop", q, t, w=-1, zdrop=zd)
        add("zdrop", q, t, w=64, zdrop=zd, flag=0x40)
    # (7) other scorings, incl. ones whose byte arithmetic wraps
    for (ma, mi, go, ge) in ((1, -2, 2, 1), (2, -4, 4, 2), (5, -4, 60, 3), (10, -9, 20, 5),
                             (5, -11, 4, 1), (1, -1, 0, 1), (5, -4, 40, 0)):
        for w in (-1, 17, 100):
            q = random_codes(rng, 250, n_frac=0.01)
            add("scoring", q, mutate(rng, q), w=w, match=ma, mismatch=mi, gapo=go, gape=ge)
    # (8) SEDEF call shapes: gap fill <=210x209, side extension 500x500, close gap <=1000^2
    for (ql, tl) in ((210, 209), (37, 3), (3, 180), (500, 500), (500, 431), (1000, 1000),
                     (1000, 640), (977, 1000)):
        q = random_codes(rng, ql)
        t = mutate(rng, q)
        t = t[:tl] if len(t) >= tl else np.concatenate([t, random_codes(rng, tl - len(t))])
        add("sedef_shape", q, t, w=-1)

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "extz2_kat.json.gz")
    blob = json.dumps(dict(source="reference extern/ksw2_extz2_sse.cc via oracle/_ref/libksw2_ref.so",
                           alphabet=ALPHA, cases=cases), separators=(",", ":")).encode()
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(blob)
    print("wrote %s: %d cases, %d bytes" % (out, len(cases), os.path.getsize(out)))


if __name__ == "__main__":
    main()
