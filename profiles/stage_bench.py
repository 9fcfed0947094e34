#!/usr/bin/env python3
"""Stage-level timing: `sedef align generate` (the product CLI, GPU provider) on a synthetic genome with planted
duplications.  usage: stage_bench.py [genome_len] [n_duplications] [runs]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostgen  # noqa: E402
from sedef_amd.host import CLI, build_host  # noqa: E402


def main():
    glen = int(sys.argv[1]) if len(sys.argv) > 1 else 40000000
    nsd = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    build_host()
    d = tempfile.mkdtemp(prefix="sdf_stage_")
    fa = os.path.join(d, "genome.fa")
    t0 = time.time()
    hostgen.make_genome(fa, seed=11, glen=glen, nsd=nsd)
    print("genome %d bp, %d planted duplications (generated in %.1fs)" % (glen, nsd, time.time() - t0), flush=True)
    for it in range(runs):
        t0 = time.time()
        r = subprocess.run([CLI, "align", "generate", "-k", "11", fa, fa + ".bed"], capture_output=True, text=True)
        dt = time.time() - t0
        tail = [ln for ln in r.stderr.replace("\r", "\n").splitlines() if "Finished" in ln or "host CPU" in ln or "driver (" in ln or "sdf_extz2_batch" in ln or "sdf_anchors" in ln or "DevBuf" in ln or "slow plan" in ln or "process:" in ln]
        print("run %d: rc=%d wall %.2fs, %d output lines\n  %s" % (it, r.returncode, dt, r.stdout.count("\n"),
                                                                  "\n  ".join(tail)), flush=True)


if __name__ == "__main__":
    main()
