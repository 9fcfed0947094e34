#!/bin/bash
# Round 6 (end): what the planner makes of the far-gap round of the chr1-sized stage (10,813 tasks) at a workspace of 8 GiB.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06pk; mkdir -p $out
d=/tmp/sdf_stage_one
python3 - > $out/gen.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
d = "/tmp/sdf_stage_one"; os.makedirs(d + "/one", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa)
PY
B=sedef_amd/bin/sedef
for ws in 8 16; do
SDF_STAGE_WS_GIB=$ws SDF_DEBUG_PLAN=1 SDF_DEBUG_TIMING=1 $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > /tmp/o.bed 2> $out/plan$ws.log
tr '\r' '\n' < $out/plan$ws.log | grep -v "Processing\|DevBuf" > $out/plan$ws.txt; rm $out/plan$ws.log
echo "== ws $ws"; awk '/n=708600/{f=1} f' $out/plan$ws.txt | awk '/n=10813/{print; exit} {print}' | cut -c1-300 | tail -4
done
