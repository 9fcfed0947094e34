import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
import hostgen
glen, nsd = int(sys.argv[1]), int(sys.argv[2])
t = time.time()
path = "/tmp/stage_genome.fa"
hostgen.make_genome(path, seed=7, glen=glen, nsd=nsd)
print("generated genome %d bp, %d SDs in %.1fs" % (glen, nsd, time.time() - t), flush=True)
for rep in range(2):
    t = time.time()
    r = subprocess.run(["sedef_amd/bin/sedef", "align", "generate", "-k", "11", path, path + ".bed"], capture_output=True, text=True)
    dt = time.time() - t
    lines = r.stdout.count("\n")
    print("rep", rep, "rc", r.returncode, "wall %.2fs" % dt, "bedpe lines", lines)
    print("\n".join(l for l in r.stderr.split("\n") if "Finished" in l or "host CPU" in l or "Error" in l))
