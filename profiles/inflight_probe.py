#!/usr/bin/env python3
"""The headline batch with ONE and with TWO calls in flight (two contexts, two host threads that take the steps in turn;
ctypes releases the GIL for the length of a call): what the host time in front of a call's first launch and the traceback /
compaction tail behind its last cost when nothing else is queued.  usage: python3 profiles/inflight_probe.py [steps]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import sedef_amd  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    n = 100000
    (pool, q_off, qlen, t_off, tlen), w = bench.synth_batch(n, 1000, seed=42), 128
    cells = int(bench.batch_cells(qlen, tlen, w).sum())
    words, q_word, t_word = bench.pack_batch(pool, q_off, qlen, t_off, tlen)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
    tasks["w"], tasks["zdrop"], tasks["flag"] = w, -1, 0
    d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
    cig_cap = 256 * n
    want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
    for depth in (1, 2, 1, 2):
        engs = [sedef_amd.Extz2Engine(0, 64 << 30) for _ in range(depth)]
        outs = [torch.empty(n * 16, dtype=torch.int32, device=dev) for _ in range(depth)]
        cigs = [torch.empty(cig_cap, dtype=torch.int32, device=dev) for _ in range(depth)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]

        def work(i, k):
            for _ in range(k):
                engs[i].align_batch_device(tasks, d_pool.data_ptr(), outs[i].data_ptr(), cigs[i].data_ptr(), cig_cap,
                                           want=want, stream=streams[i].cuda_stream)

        for i in range(depth):
            work(i, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        thr = [threading.Thread(target=work, args=(i, steps // depth)) for i in range(depth)]
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k = steps // depth * depth
        print("calls in flight %d: %d steps %.2f ms per step  %.1f Gcell/s" % (depth, k, dt / k * 1e3, cells * k / dt / 1e9))
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print("   results of the contexts equal:", same)
        del engs


if __name__ == "__main__":
    main()
