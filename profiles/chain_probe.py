#!/usr/bin/env python3
"""Chained strips (extz2_strip.hip) alone: n full-band tasks of q x t, timed as one batch.  usage: chain_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402

eng = sedef_amd.Extz2Engine(0, 64 << 30)
dev = torch.device("cuda", 0)
rng = np.random.Generator(np.random.MT19937(5))
for n, q, t in ((2, 6000, 6000), (16, 6000, 6000), (128, 6000, 6000), (2, 3000, 3000), (64, 3000, 3000), (1024, 3000, 3000),
                (2, 1000, 1000), (2048, 1000, 1000), (20000, 1000, 1000)):
    ql = np.full(n, q)
    shapes_bench.run("%d x %d x %d" % (n, q, t), bench.synth_ragged(rng, ql, tlens=np.full(n, t)), -1, eng, dev)
