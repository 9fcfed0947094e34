#!/bin/bash
# Round 6: the far-gap round of the chr1-sized stage (10,813 tasks, two chunks of ~7 ms stripe chains): workspace sizes and
# stripe widths.   bash profiles/r06_stage_dp2.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06dp2; mkdir -p $out
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
run() {  # label, env...
  label=$1; shift
  for i in 1 2 3; do
    env "$@" SDF_DEBUG_TIMING=1 $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_$label.bed 2> $out/$label.$i.log
    tr '\r' '\n' < $out/$label.$i.log | grep -v "Processing\|DevBuf" > $out/$label.$i.txt; rm $out/$label.$i.log
    echo "$label: $(grep -o 'process:.*' $out/$label.$i.txt) | $(grep 'n=10813' $out/$label.$i.txt | grep -o 'h2d+device.*') sha $(sha256sum < $d/out_$label.bed | cut -c1-12)"
  done
}
run base SDF_X=0
run ws16 SDF_STAGE_WS_GIB=16
run ws24 SDF_STAGE_WS_GIB=24
run ws16n1 SDF_STAGE_WS_GIB=16 SDF_STRIPE_NREG=1
run n1 SDF_STRIPE_NREG=1
run chain SDF_STAGE_WS_GIB=16 SDF_CHAIN_MIN=64
