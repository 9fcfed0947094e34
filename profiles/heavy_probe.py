#!/usr/bin/env python3
"""The heavy tasks of the hg19-shaped mixture alone (and the ordinary ones alone): how long each part takes when it
has the GPU to itself.  usage: heavy_probe.py [n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda", 0)
(pool, q_off, qlen, t_off, tlen), w = bench.synth_hg19_mixture_fast(n, seed=404, big=6000)
eng = sedef_amd.Extz2Engine(0, 64 << 30)
for name, sel in (("stripe tasks (>= 1200)", qlen >= 1200), ("600..1000", (qlen >= 600) & (qlen < 1200)),
                  ("500 x ~500", qlen == 500), ("long + 500 x ~500", qlen >= 500), ("ordinary (< 500)", qlen < 500), ("all", qlen > 0)):
    idx = np.flatnonzero(sel)
    b = (pool, q_off[idx], qlen[idx], t_off[idx], tlen[idx])
    shapes_bench.run(name, b, w, eng, dev, steps=3)
