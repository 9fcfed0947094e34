#!/usr/bin/env python3
"""Throughput of the batched DP path on the other BASELINE shapes (parity-test cases, not the bench line):
configs[1] inputs with w=-1 (SEDEF's real mode), the hg19-shaped task mixture (configs[3]) and the mm8-like mixed-band
batch (configs[4]).  Inputs resident in HBM, same timing as bench.py (whole call: planning + DP + traceback + compaction).
usage: python3 profiles/shapes_bench.py [n_hg19] [n_mm8] [n_mm8_throughput]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sedef_amd  # noqa: E402


def run(name, batch, w, eng, dev, steps=3):
    pool, q_off, qlen, t_off, tlen = batch
    n = len(qlen)
    words, q_word, t_word = bench.pack_batch(pool, q_off, qlen, t_off, tlen)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
    tasks["w"], tasks["zdrop"] = w, -1
    ws = np.broadcast_to(np.asarray(w, np.int32), (n,))
    cells_task = bench.batch_cells(qlen, tlen, w)
    cells = int(cells_task.sum())
    d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
    d_out = torch.empty(n * 16, dtype=torch.int32, device=dev)
    cap = int((qlen.astype(np.int64) + tlen + 2).sum())
    d_cig = torch.empty(cap, dtype=torch.int32, device=dev)
    want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
    if os.environ.get("SHAPES_SCORE_ONLY") == "1":  # (probe: the DP without its direction flags and traceback)
        want = sedef_amd.extz2.WANT_SCORE
    eng.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cap, want=want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cap, want=want)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = d_out.cpu().numpy().view(sedef_amd.RESULT_DTYPE)
    alg = bench.algorithmic_bytes(qlen, tlen, cells_task, res["n_cigar"])  # SURVEY 8(d), whole call
    print("%-34s tasks %8d  cells %.3e  %8.2f ms/step  %8.1f Gcell/s  hbm_frac %.4f  paired %d  launches %d  dp %.2f tb %.2f plan %.2f ms"
          % (name, n, cells, dt * 1e3, cells / dt / 1e9, alg / dt / 1e9 / bench.HBM_PEAK_GBS, eng.last_paired(),
             eng.last_launches(), eng.last_ms(0), eng.last_ms(1), eng.last_ms(4)), flush=True)


def main():
    n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    n5 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    n5big = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda", 0)
    eng = sedef_amd.Extz2Engine(0, 64 << 30)
    run("configs[1] w=128 (bench line)", bench.synth_batch(100000, 1000, 42), 128, eng, dev)
    run("configs[1] inputs, w=-1 (20k tasks)", bench.synth_batch(20000, 1000, 42), -1, eng, dev)
    b4, w4 = bench.synth_hg19_mixture_fast(n4, seed=404, big=6000)
    run("configs[3] hg19 task mixture w=-1", b4, w4, eng, dev)
    b5, w5 = bench.synth_mm8_mixture(n5, seed=505)
    run("configs[4] mm8 mixed bands 64..512", b5, w5, eng, dev)
    if n5big:  # the same distribution at a size that fills the device: throughput, where the line above is tail latency
        b6, w6 = bench.synth_mm8_mixture_fast(n5big, seed=505)
        run("configs[4] mm8, %d tasks" % n5big, b6, w6, eng, dev, steps=2)


if __name__ == "__main__":
    main()
