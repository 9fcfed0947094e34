#!/bin/bash
# Round 6: run-to-run variance of the chr1-sized bucket's stage clock, with the phase timeline of every run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06var; mkdir -p $out
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
sync; sleep 2
B=sedef_amd/bin/sedef
run() {  # label, env...
  label=$1; shift
  for i in 1 2 3 4 5; do
    env "$@" SDF_DEBUG_TIMING=1 $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_$label.bed 2> $out/$label.$i.log
    tr '\r' '\n' < $out/$label.$i.log | grep -v "Processing\|DevBuf" > $out/$label.$i.txt; rm $out/$label.$i.log
    echo "$label: $(grep -o 'process:.*' $out/$label.$i.txt) $(grep -o 'sdf_reserve.*' $out/$label.$i.txt) $(grep -o 'sdf_pool_host.*' $out/$label.$i.txt) | fetched $(grep 'sequences fetched' $out/$label.$i.txt | grep -o '[0-9.]* ms') anchors $(grep 'anchors done' $out/$label.$i.txt | grep -o '[0-9.]* ms') chained $(grep -m1 'requests collected' $out/$label.$i.txt | grep -o '[0-9.]* ms') dp1 $(grep -m1 'DP round done' $out/$label.$i.txt | grep -o '[0-9.]* ms')"
  done
}
run huge SDF_X=0
run plain SDF_POOL_PLAIN=1
run huge2 SDF_X=0
