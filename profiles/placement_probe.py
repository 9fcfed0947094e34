#!/usr/bin/env python3
"""Where the dispatcher puts the chain wavefronts of the hg19 mixture's heavy chunk: wavefronts per SIMD (sdf_debug_placement).
usage: placement_probe.py [n_tasks]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
eng = sedef_amd.Extz2Engine(0, 64 << 30)
lib = eng.lib
lib.sdf_debug_placement.argtypes = [C.c_void_p, C.c_void_p]
lib.sdf_debug_placement(eng.ctx, None)
dev = torch.device("cuda", 0)
b, w = bench.synth_hg19_mixture_fast(n, seed=404, big=6000)
shapes_bench.run("hg19", b, w, eng, dev, steps=1)   # one warm-up + one timed call: two launches' worth of counts
out = np.zeros(4096, np.uint32)
lib.sdf_debug_placement(eng.ctx, out.ctypes.data)
used = out[out > 0]
print("chain wavefronts counted %d on %d SIMDs: per SIMD min %d  mean %.2f  max %d" % (out.sum(), len(used), used.min(), used.mean(), used.max()))
per_xcd = out.reshape(8, 512).sum(1)
print("per XCD:", per_xcd.tolist())
per_cu = out.reshape(-1, 4).sum(1)
pc = per_cu[per_cu > 0]
print("per CU (%d CUs): min %d mean %.1f max %d" % (len(pc), pc.min(), pc.mean(), pc.max()))
print("histogram of wavefronts per SIMD:", np.bincount(used).tolist())
