# usage: bash profiles/stage_trace.sh [genome_len] [n_dup]  -- kernel trace of one `sedef align generate` run (one lane)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=${1:-100000000}; N=${2:-10000}
python3 - <<PY
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hostgen
from sedef_amd.host import build_host
build_host()
hostgen.make_genome("/tmp/stage_trace.fa", seed=11, glen=$G, nsd=$N)
PY
export SDF_LANES=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stage_trace -o run -- sedef_amd/bin/sedef align generate -k 11 /tmp/stage_trace.fa /tmp/stage_trace.fa.bed > /tmp/stage_trace.out 2> gpurun_out/stage_trace.log
tr '\r' '\n' < gpurun_out/stage_trace.log | grep "Finished" 
python3 profiles/timeline.py gpurun_out/stage_trace 100000 > gpurun_out/stage_timeline.txt
python3 - <<'PY'
import re, collections
rows = [l.split() for l in open("gpurun_out/stage_timeline.txt")]
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    name = " ".join(r[5:]); tot[name] += float(r[2]); cnt[name] += 1
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:25]:
    print("%9.2f ms  %5d  %s" % (v, cnt[k], k))
PY
