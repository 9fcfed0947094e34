#!/usr/bin/env python3
"""Times sdf_chain_batch (GPU: one wavefront per pair with everything in LDS for pairs of up to ~400 anchors, one thread per
pair beyond; SDF_CHAIN_THREADS=1: one thread per pair throughout) against the host chain_anchors on the same anchor sets.  The
GPU figure is the whole call through the Python binding (list handling, uploads, kernels, downloads): kernel times come from
`rocprofv3 --kernel-trace --stats` of this script.
usage: chain_bench.py [npairs] [seqlen]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hostgen  # noqa: E402
import sedef_amd  # noqa: E402
import sedef_amd.host as host  # noqa: E402


def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seqlen = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    rng = np.random.default_rng(5)
    base = []
    for it in range(32):
        q = hostgen.rseq(rng, int(seqlen * (0.3 + 1.4 * rng.random())))
        r = hostgen.mut(rng, q, 0.02 + rng.random() * 0.12)
        base.append(np.array(host.anchors(q, r, 11), np.int32).reshape(-1, 4))
    cases = [base[i % len(base)] for i in range(npairs)]
    tot = sum(len(a) for a in cases)
    eng = sedef_amd.Extz2Engine(0)
    eng.chain_batch(cases[:64])
    t0 = time.perf_counter()
    got = eng.chain_batch(cases)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    exp = [host.chain_raw(a) for a in base]
    t_host = (time.perf_counter() - t0) * npairs / len(base)
    for i in range(len(base)):
        assert np.array_equal(got[i][0], exp[i][0]) and np.array_equal(got[i][1], exp[i][1])
    print("pairs %d anchors %d (max %d per pair): gpu %.1f ms, host 1 thread %.1f ms (%.2f us/anchor)" % (
        npairs, tot, max(len(a) for a in base), t_gpu * 1e3, t_host * 1e3, t_host * 1e6 / tot))


if __name__ == "__main__":
    main()
