#!/usr/bin/env python3
"""One warm-up and one timed call of a mixture batch, for `rocprofv3 --kernel-trace` + profiles/timeline.py.
usage: mix_probe.py mm8|hg19 [n] [workspace GiB]"""
import os
import sys

# (a batch of banded tasks of all lengths runs its launches side by side on up to sixteen streams, sdf_launch.hip: the
# runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues -- bench.py's default of 8 is for the headline's four)
if len(sys.argv) > 2 and sys.argv[1] == "mm8" and int(sys.argv[2]) >= 20000:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "mm8"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (3000 if which == "mm8" else 300000)
# (the mm8-like batch of 100,000 tasks holds 131 GB of direction flags: at 128 GiB it runs as three chunks, at 64 GiB as six --
# 201 against 252 ms, profiles/r04_shapes.txt.  Round 5: the library's default, `sdf_create(device, 0)`, is half of the free HBM,
# and that is what this probe runs with unless a figure is given)
ws_gib = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng = sedef_amd.Extz2Engine(0, ws_gib << 30)
dev = torch.device("cuda", 0)
if which == "mm8":
    b, w = bench.synth_mm8_mixture(n, seed=505) if n <= 5000 else bench.synth_mm8_mixture_fast(n, seed=505)
else:
    b, w = bench.synth_hg19_mixture_fast(n, seed=404, big=6000)
shapes_bench.run(which, b, w, eng, dev, steps=1)
