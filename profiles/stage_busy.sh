# usage: bash profiles/stage_busy.sh [genome_len] [n_dup]  -- `sedef align generate` (product CLI, default lanes) under
# rocprofv3 --kernel-trace: stage wall time, union of the kernel intervals (GPU busy), host thread time of the stage
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=${1:-100000000}; N=${2:-40000}
python3 - <<PY
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hostgen
from sedef_amd.host import build_host
build_host()
hostgen.make_genome("/tmp/stage_busy.fa", seed=11, glen=$G, nsd=$N)
PY
export GPU_MAX_HW_QUEUES=8
sedef_amd/bin/sedef align generate -k 11 /tmp/stage_busy.fa /tmp/stage_busy.fa.bed > /tmp/stage_busy.out 2> /tmp/stage_busy.warm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stage_busy -o run -- sedef_amd/bin/sedef align generate -k 11 /tmp/stage_busy.fa /tmp/stage_busy.fa.bed > /tmp/stage_busy.out 2> gpurun_out/stage_busy.log
tr '\r' '\n' < /tmp/stage_busy.warm | grep -A2 "Finished" | sed 's/^/untraced: /'
tr '\r' '\n' < gpurun_out/stage_busy.log | grep -A2 "Finished" | sed 's/^/traced:   /'
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/stage_busy/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-36:]))
rows.sort()
busy, end = 0, -1
for s, e, _ in rows:
    if s > end:
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
span = rows[-1][1] - rows[0][0]
print("kernels %d; first to last kernel %.3f s; union of kernel intervals (GPU busy) %.3f s = %.0f %% of that span"
      % (len(rows), span / 1e9, busy / 1e9, 100.0 * busy / span))
import collections
tot = collections.defaultdict(float)
for s, e, k in rows:
    tot[k] += (e - s) / 1e6
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:10]:
    print("  %9.2f ms  %s" % (v, k))
PY
rm -rf gpurun_out/stage_busy
