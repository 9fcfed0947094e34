"""GPU fuzz of the whole stage (not collected by pytest: run by hand on a GPU box, e.g.  SEED=1 ROUNDS=12 python tests/fuzz/fuzz_stage.py).
Every round makes a genome with planted duplications (tests/hostgen.py), runs `align generate` three ways -- the host pipeline
with the reference kernel as its DP (oracle/_ref through the test hook), the same pipeline on the GPU provider through the C ABI,
the product CLI on one and on three lanes with small super-batches -- and compares the four outputs byte for byte."""
import ctypes as C, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hostgen
from sedef_amd import host
from sedef_amd.host import CLI, build_host
from oracle.binding import build_reference
build_host()
ref = C.CDLL(build_reference())
hook = C.cast(ref.ref_extz2_hook, C.c_void_p)
seed0 = int(os.environ.get("SEED", "1")); rounds = int(os.environ.get("ROUNDS", "8"))
bad = 0; pairs = 0; lines = 0; t0 = time.time()
for rd in range(rounds):
    rng = np.random.default_rng(seed0 * 7919 + rd)
    d = tempfile.mkdtemp(prefix="sdf_fz_")
    fa = os.path.join(d, "g.fa")
    glen = int(rng.integers(100000, int(os.environ.get("GLEN_MAX", "2000000"))))
    nsd = int(rng.integers(6, int(os.environ.get("NSD_MAX", "160"))))
    hostgen.make_genome(fa, seed=int(rng.integers(1, 1 << 30)), glen=glen, nsd=nsd)
    cpu, gpu = os.path.join(d, "cpu.bed"), os.path.join(d, "gpu.bed")
    st = host.generate(fa, fa + ".bed", 11, cpu, test_dp=hook)
    host.generate(fa, fa + ".bed", 11, gpu)
    want = open(cpu).read()
    outs = [open(gpu).read()]
    for env in ({"SDF_LANES": "1"}, {"SDF_LANES": "3", "SDF_SUPER_BATCH": str(max(1, nsd // 7))}):
        r = subprocess.run([CLI, "align", "generate", "-k", "11", fa, fa + ".bed"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        outs.append(r.stdout if r.returncode == 0 else "rc %d: %s" % (r.returncode, r.stderr[-300:]))
    ok = all(o == want for o in outs)
    pairs += st[0]; lines += want.count("\n")
    if not ok:
        bad += 1
        print("BAD round %d (genome %d bp, %d duplications, kept in %s)" % (rd, glen, nsd, d), [o == want for o in outs], flush=True)
    else:
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        os.rmdir(d)
print("fuzz_stage: %d genomes, %d seed pairs, %d output lines, %d bad, %.0f s" % (rounds, pairs, lines, bad, time.time() - t0))
