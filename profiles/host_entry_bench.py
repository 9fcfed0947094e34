#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (sdf_extz2_batch) on BASELINE configs[1]: sequences as byte
codes in host memory, results and CIGARs back in host memory.  Never bench.py's `value` (that one has the inputs
resident in HBM); reported in DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sedef_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 1000, 42)
tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
tasks["w"], tasks["zdrop"] = 128, -1
cells = sum(sedef_amd.band_cells(int(a), int(b), 128) for a, b in zip(qlen[:200], tlen[:200])) / 200 * n
eng = sedef_amd.Extz2Engine(0, 64 << 30)
want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
eng.align_batch(tasks, pool, want=want)
t0 = time.perf_counter()
for _ in range(3):
    res, cig = eng.align_batch(tasks, pool, want=want)
dt = (time.perf_counter() - t0) / 3
print("host entry: %d tasks, %.1f ms per call (python wrapper included), ~%.0f Gcell/s" % (n, dt * 1e3, cells / dt / 1e9))
