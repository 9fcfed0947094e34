#!/bin/bash
# Round 6: the chr1-sized bucket (one bucket file, one process) with the lanes / super-batch sizes given in the environment:
# what overlap of super-batches buys on the stage clock.   bash profiles/r06_stage_lanes.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06lanes; mkdir -p $out
d=/tmp/sdf_stage_one
python3 - > $out/gen.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
from sedef_amd.host import build_host
build_host()
d = "/tmp/sdf_stage_one"; os.makedirs(d + "/one", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa)
print(nseeds)
PY
B=sedef_amd/bin/sedef
now() { date +%s.%N; }
run() {  # label, env...
  label=$1; shift
  for i in 1 2 3; do
    t0=$(now); env "$@" $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_$label.bed 2> $out/$label.$i.log; t1=$(now)
    echo "$label: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(tr '\r' '\n' < $out/$label.$i.log | grep -o 'Finished BED.*' | grep -o 'in [0-9.]*s'); sha $(sha256sum < $d/out_$label.bed | cut -c1-12)"
  done
}
run base SDF_X=0
run l2 SDF_LANES=2
run l2s512 SDF_LANES=2 SDF_SUPER_BATCH=512
run l2s256 SDF_LANES=2 SDF_SUPER_BATCH=256
run l3s256 SDF_LANES=3 SDF_SUPER_BATCH=256
run l3s128 SDF_LANES=3 SDF_SUPER_BATCH=128
run l4s128 SDF_LANES=4 SDF_SUPER_BATCH=128
