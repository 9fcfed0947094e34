#!/usr/bin/env python3
"""hg19-shaped task mixture (BASELINE configs[3]) under different shares of the CUs for its heavy tasks
(SDF_HEAVY_CU_FRAC, read by sdf_create).  usage: hg19_sweep.py [n] [frac ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
fracs = [float(x) for x in sys.argv[2:]] or [0.0, 0.5, 0.625, 0.75, 0.875]
dev = torch.device("cuda", 0)
b, w = bench.synth_hg19_mixture(n, seed=404, big=6000)
for f in fracs:
    os.environ["SDF_HEAVY_CU_FRAC"] = str(f)
    eng = sedef_amd.Extz2Engine(0, 64 << 30)
    shapes_bench.run("hg19 mixture, heavy CU share %.3f" % f, b, w, eng, dev, steps=3)
    eng.close()
