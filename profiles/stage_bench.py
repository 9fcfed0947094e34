#!/usr/bin/env python3
"""Stage-level timing of `sedef align generate` on a synthetic genome with planted duplications.

  stage_bench.py [genome_len] [n_duplications] [runs]     the product CLI (GPU provider) on one bucket (round 1 / 2 genomes)
  stage_bench.py --chr1 [runs]                            BASELINE configs[2] at size: tests/hostgen.py: make_chr1_genome
                                                          (249 Mb, 1-100 kb copies at 2-25 %) -> align bucket (4) -> generate
  ... --one-bucket                                        (with --chr1) all seed pairs in ONE bucket file: one process
  ... --cpu                                               the same host pipeline with the REFERENCE kernel as the DP
                                                          (oracle/_ref: ksw_extz2_sse behind the library's test hook, one
                                                          task stream per usable host core) instead of the GPU: the
                                                          stage-level CPU baseline of SURVEY 8(d)

The reference itself runs one single-threaded process per bucket (sedef.sh:187-190); its `align generate` cannot be built
here (Boost), so the CPU leg is this repository's host code -- anchors, chaining, refinement on all usable cores -- with the
reference's kernel doing every DP call."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostgen  # noqa: E402
from sedef_amd import host  # noqa: E402
from sedef_amd.host import CLI, build_host  # noqa: E402

KEEP = ("Finished", "host CPU", "driver (", "sdf_extz2_batch", "sdf_anchors", "DevBuf", "slow plan", "process:", "[stage ", "sdf_create", "[class ", "[batch ", "[plan", "[sdf_reserve", "[chunk ", "[lane ")


def cores():
    import bench
    return bench.effective_cores()


def cpu_hook():
    from oracle.binding import build_reference
    lib = C.CDLL(build_reference())
    return C.cast(lib.ref_extz2_hook, C.c_void_p), lib


def run_cli(fa, bed):
    t0 = time.time()
    r = subprocess.run([CLI, "align", "generate", "-k", "11", fa, bed], capture_output=True, text=True)
    dt = time.time() - t0
    tail = [ln for ln in r.stderr.replace("\r", "\n").splitlines() if any(k in ln for k in KEEP)]
    return r.returncode, dt, r.stdout.count("\n"), tail


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    chr1, cpu = "--chr1" in sys.argv, "--cpu" in sys.argv
    build_host()
    d = tempfile.mkdtemp(prefix="sdf_stage_")
    fa = os.path.join(d, "genome.fa")
    t0 = time.time()
    if chr1:
        runs = int(argv[0]) if argv else 2
        genome, nseeds = hostgen.make_chr1_genome(fa)
        out = os.path.join(d, "buckets")
        os.makedirs(out)
        host.bucket(fa + ".seeds.bed", 1 if "--one-bucket" in sys.argv else 4, out, fa)
        beds = [os.path.join(out, f) for f in sorted(os.listdir(out))]
        print("chr1-sized genome (%d + %d bp), %d seed pairs in %d buckets (generated in %.1fs)" % (
            len(genome["chr1"]), len(genome["chr1b"]), nseeds, len(beds), time.time() - t0), flush=True)
        del genome
    else:
        glen = int(argv[0]) if len(argv) > 0 else 40000000
        nsd = int(argv[1]) if len(argv) > 1 else 2000
        runs = int(argv[2]) if len(argv) > 2 else 2
        hostgen.make_genome(fa, seed=11, glen=glen, nsd=nsd)
        beds = [fa + ".bed"]
        print("genome %d bp, %d planted duplications (generated in %.1fs)" % (glen, nsd, time.time() - t0), flush=True)
    hook = cpu_hook() if cpu else None
    for it in range(runs):
        total, lines, tasks, cells, stage_s = 0.0, 0, 0, 0, 0.0
        for bed in beds:
            if cpu:
                t0 = time.time()
                st = host.generate(fa, bed, 11, os.path.join(d, "cpu.bed"), test_dp=hook[0])
                dt = time.time() - t0
                lines, tasks, cells = lines + st[0], tasks + st[2], cells + st[3]
                print("  run %d %s: CPU leg wall %.2fs, %d lines, %d DP tasks, %.3e cells" % (it, os.path.basename(bed), dt, st[0], st[2], st[3]),
                      flush=True)
            else:
                rc, dt, nl, tail = run_cli(fa, bed)
                lines += nl
                own = [ln for ln in tail if "Finished" in ln]
                if own:  # the stage's own clock (src/align_main.cc:335-336 prints the same line): without process start-up
                    stage_s += float(own[0].split(" in ")[1].split("s")[0])
                print("  run %d %s: rc=%d wall %.2fs, %d output lines\n    %s" % (it, os.path.basename(bed), rc, dt, nl, "\n    ".join(tail)),
                      flush=True)
            total += dt
        print("run %d: %s, all buckets one after the other: %.2fs wall, %d output lines%s" % (
            it, "CPU leg (reference kernel on %d usable cores of %d)" % (cores(), os.cpu_count()) if cpu else "GPU path (product CLI)",
            total, lines, (", %d DP tasks, %.3e cells" % (tasks, cells)) if cpu else
            (" (%.2fs on the stage's own clock: without process start-up and HIP initialisation)" % stage_s)), flush=True)


if __name__ == "__main__":
    main()
