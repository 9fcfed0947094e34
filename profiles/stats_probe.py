"""Throughput of sdf_stats_columns_device on synthetic SD-shaped alignments resident in HBM.
python profiles/stats_probe.py [n_alignments] [mean_len] -> GB/s of algorithmic bytes (a_len + b_len + 4 n_cigar + 64)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sedef_amd  # noqa: E402
from sedef_amd.extz2 import STATS_COLS_DTYPE, STATS_TASK_DTYPE  # noqa: E402


def build(n, mean_len, seed=1):
    rng = np.random.default_rng(seed)
    lens = np.maximum(200, rng.exponential(mean_len, n)).astype(np.int64)
    # run structure of a ~10 % divergent alignment: M runs of ~40 columns, a short gap between them
    tasks = np.zeros(n, STATS_TASK_DTYPE)
    cig_parts, off, coff = [], 0, 0
    for k in range(n):
        nrun = max(1, int(lens[k] // 45))
        m = rng.integers(1, 80, nrun).astype(np.uint32)
        g = rng.integers(1, 6, nrun).astype(np.uint32)
        op = rng.integers(1, 3, nrun).astype(np.uint32)
        cg = np.empty(2 * nrun, np.uint32)
        cg[0::2] = m << 4
        cg[1::2] = (g << 4) | op
        na = int(m.sum() + g[op == 1].sum())
        nb = int(m.sum() + g[op == 2].sum())
        tasks[k] = (off, off + na, na, nb, coff, len(cg), 0)
        off += na + nb
        coff += len(cg)
        cig_parts.append(cg)
    pool = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), off)
    return tasks, pool, np.concatenate(cig_parts)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    mean_len = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    tasks, pool, cig = build(n, mean_len)
    eng = sedef_amd.Extz2Engine(0)
    dev = torch.device("cuda:0")
    d_tasks = torch.from_numpy(tasks.view(np.uint8)).to(dev)
    d_pool = torch.from_numpy(pool).to(dev)
    d_cig = torch.from_numpy(cig.view(np.int32)).to(dev)
    d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    bytes_alg = int(tasks["a_len"].sum() + tasks["b_len"].sum()) + 4 * len(cig) + 64 * n
    st = torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(st):
        for _ in range(3):
            eng.stats_columns_device(d_tasks.data_ptr(), n, d_pool.data_ptr(), d_cig.data_ptr(), d_out.data_ptr(),
                                     st.cuda_stream)
        ev[0].record(st)
        reps = 20
        for _ in range(reps):
            eng.stats_columns_device(d_tasks.data_ptr(), n, d_pool.data_ptr(), d_cig.data_ptr(), d_out.data_ptr(),
                                     st.cuda_stream)
        ev[1].record(st)
    st.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    out = d_out.cpu().numpy().view(STATS_COLS_DTYPE)
    cols = int(out["span"].astype(np.int64).sum())
    assert int(out["flags"].sum()) == 0
    print("stats columns: %d alignments, %.1f M columns, %.1f MB algorithmic: %.3f ms per launch = %.0f GB/s "
          "(%.3f of 8 TB/s), %.1f Gcolumn/s" % (n, cols / 1e6, bytes_alg / 1e6, ms, bytes_alg / ms / 1e6,
                                              bytes_alg / ms / 1e6 / 8000, cols / ms / 1e6))


if __name__ == "__main__":
    main()
