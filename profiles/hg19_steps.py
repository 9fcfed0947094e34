#!/usr/bin/env python3
"""The hg19-shaped mixture (BASELINE configs[3]) step by step: whole-call time, DP interval and planning time of every step.
usage: python3 profiles/hg19_steps.py [n_tasks] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sedef_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
eng = sedef_amd.Extz2Engine(0, 64 << 30)
(pool, q_off, qlen, t_off, tlen), w = bench.synth_hg19_mixture_fast(n, seed=404, big=6000)
words, q_word, t_word = bench.pack_batch(pool, q_off, qlen, t_off, tlen)
tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
tasks["w"], tasks["zdrop"] = w, -1
d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
d_out = torch.empty(n * 16, dtype=torch.int32, device=dev)
cap = int((qlen.astype(np.int64) + tlen + 2).sum())
d_cig = torch.empty(cap, dtype=torch.int32, device=dev)
want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
out = []
for s in range(steps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cap, want=want)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if s:
        out.append(dt)
        print("step %2d  %6.2f ms   dp %5.2f  tb %4.2f  first launch after %4.2f ms  reran %d" % (
            s, dt, eng.last_ms(0), eng.last_ms(1), eng.last_ms(4), eng.last_reran()), flush=True)
print("median %.2f ms, min %.2f, max %.2f" % (float(np.median(out)), min(out), max(out)))
