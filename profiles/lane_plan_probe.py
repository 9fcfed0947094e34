#!/usr/bin/env python3
"""The hg19-shaped mixture (BASELINE configs[3]) through contexts that differ in how the lane path is planned on the device
(round 4: hipCUB radix sort + scans; round 5: counting sort over the key bins) and in the priority of its stream: same batch,
same process, alternating; every record and CIGAR word compared.  usage: lane_plan_probe.py [n]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shapes_bench  # noqa: E402
from shapes_bench import bench, sedef_amd  # noqa: E402


def engine(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return sedef_amd.Extz2Engine(0, 48 << 30)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
(pool, q_off, qlen, t_off, tlen), w = bench.synth_hg19_mixture_fast(n, seed=404, big=6000)
words, q_word, t_word = bench.pack_batch(pool, q_off, qlen, t_off, tlen)
tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
tasks["w"], tasks["zdrop"] = w, -1
cells = int(bench.batch_cells(qlen, tlen, w).sum())
dev = torch.device("cuda", 0)
d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
d_out = torch.empty(n * 16, dtype=torch.int32, device=dev)
cap = int((qlen.astype(np.int64) + tlen + 2).sum())
d_cig = torch.empty(cap, dtype=torch.int32, device=dev)
want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
configs = [("round 4: hipCUB sort + scans, lane stream like the others", dict(SDF_LANE_PLAN="sort", SDF_LANE_PRIO=0)),
           ("hipCUB sort + scans, lane stream on the highest priority", dict(SDF_LANE_PLAN="sort")),
           ("counting sort over the key bins, highest priority", {}),
           ("counting sort, lane stream like the others", dict(SDF_LANE_PRIO=0))]
engs = [(name, engine(**env)) for name, env in configs]
ref = None
for name, e in engs:  # warm-up + equality of everything that comes back
    used = e.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cap, want=want)
    torch.cuda.synchronize()
    got = (d_out.cpu().numpy().copy(), d_cig[:used].cpu().numpy().copy())
    if ref is None:
        ref = got
    else:
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]), name
    assert e.last_reran() == 0, name
for rnd in range(3):
    for name, e in engs:
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cap, want=want)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("round %d  %-44s %s ms  best %.2f  (%.0f Gcell/s)  dp %.2f tb %.2f plan %.2f launches %d" % (
            rnd, name, " ".join("%.2f" % t for t in ts), min(ts), cells / min(ts) / 1e6, e.last_ms(0), e.last_ms(1), e.last_ms(4),
            e.last_launches()), flush=True)
