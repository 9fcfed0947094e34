#!/bin/bash
# Round 6: the resident-sequence DP path -- its GPU tests, the stage tests, and the chr1-sized bucket's stage clock / timeline.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06check; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_extz2.py -m gpu -x -q -k "resident or brief" > $out/t1.txt 2>&1; tail -3 $out/t1.txt
timeout 900 python3 -m pytest tests/test_stage_pairs.py tests/test_stage_scale.py tests/test_dropin.py tests/test_host_pipeline.py tests/test_pinning.py -m gpu -x -q > $out/t2.txt 2>&1; tail -3 $out/t2.txt
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
now() { date +%s.%N; }
run() {  # label, env...
  label=$1; shift
  for i in 1 2 3; do
    t0=$(now); env "$@" $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_$label.bed 2> $out/$label.$i.log; t1=$(now)
    echo "$label: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(tr '\r' '\n' < $out/$label.$i.log | grep -o 'Finished BED.*' | grep -o 'in [0-9.]*s'); sha $(sha256sum < $d/out_$label.bed | cut -c1-12)"
  done
}
run base SDF_X=0
run nores SDF_RESIDENT_DP=0
run l2s512 SDF_LANES=2 SDF_SUPER_BATCH=512
run l3s256 SDF_LANES=3 SDF_SUPER_BATCH=256
SDF_DEBUG_TIMING=1 $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > /tmp/o.bed 2> $out/stage_dbg.log
tr '\r' '\n' < $out/stage_dbg.log | grep -v "Processing\|DevBuf" > $out/stage_dbg.txt; rm $out/stage_dbg.log
cat $out/stage_dbg.txt | cut -c1-250
