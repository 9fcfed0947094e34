#!/usr/bin/env python3
"""n full-band tasks of L x L in one batch, whole call (planner at its defaults or with SDF_CHAIN_MIN / SDF_NO_STRIP set):
usage: python3 profiles/fullband_probe.py <n> <L>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import bench  # noqa: E402
import sedef_amd  # noqa: E402
import shapes_bench  # noqa: E402

n, L = int(sys.argv[1]), int(sys.argv[2])
eng = sedef_amd.Extz2Engine(0, 32 << 30)
shapes_bench.run("%d x %d^2 full band" % (n, L), bench.synth_batch(n, L, 7), -1, eng, torch.device("cuda", 0))
