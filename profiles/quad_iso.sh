#!/bin/bash
# isolated launch time of the headline batch's DP kernels: SDF_PIPELINE=0, one call at a time, rocprofv3 --stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp SDF_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qiso -o q -- python3 bench.py --inflight 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pcie-pass > gpurun_out/qiso.log 2>&1
f=$(find gpurun_out/qiso -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "extz2" in r["Name"]:
        print("%-60s calls %s avg %.3f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf gpurun_out/qiso
