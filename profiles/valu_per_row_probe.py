#!/usr/bin/env python3
"""One isolated launch (SDF_PIPELINE=0) of n tasks of L x ~L at w = 128, for `rocprofv3 --pmc SQ_INSTS_VALU`: two lengths give the
headline kernel's VALU instructions per steady row (the slope) and what its first and last rows cost beyond that (the intercept).
usage: valu_per_row_probe.py L [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, sedef_amd
L = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
eng = sedef_amd.Extz2Engine(0, 32 << 30, config=dict(SDF_PIPELINE=0))
pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, L, seed=42)
pairs = [(pool[q_off[i]:q_off[i] + qlen[i]], pool[t_off[i]:t_off[i] + tlen[i]]) for i in range(n)]
res, cig = eng.align_pairs(pairs, w=128, want=3)
print("L", L, "n", n, "rows", int((qlen.astype(np.int64) + tlen - 1).sum()), "paired", eng.last_paired(), "launches", eng.last_launches())
