#!/usr/bin/env python3
"""The lane kernel (extz2_lane.hip) alone: batches of the hg19 mixture's two small classes, without its heavy tasks.
usage: lane_probe.py [n]   (SDF_NO_LANE=1 in the environment: the same batches on the window kernels)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
eng = sedef_amd.Extz2Engine(0, 64 << 30)
dev = torch.device("cuda", 0)
rng = np.random.Generator(np.random.MT19937(11))
a1, b1 = rng.integers(1, 11, n), rng.integers(1, 11, n)
a2 = rng.integers(5, 101, n)
b2 = np.clip(a2 + rng.integers(-20, 21, n), 1, 209)
shapes_bench.run("tiny: 1..10 x 1..10", bench.synth_ragged(rng, a1, tlens=b1), -1, eng, dev)
shapes_bench.run("small: 5..100 x +-20", bench.synth_ragged(rng, a2, tlens=b2), -1, eng, dev)
for lo, hi in ((5, 32), (33, 64), (65, 100)):
    a3 = rng.integers(lo, hi + 1, n)
    shapes_bench.run("small: %d..%d square" % (lo, hi), bench.synth_ragged(rng, a3, tlens=a3), -1, eng, dev)
print("lane tasks of the last call:", eng.last_lane_tasks())
