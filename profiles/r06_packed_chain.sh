#!/bin/bash
# Round 6 (end): the chained strips' flags packed (four columns a lane: one record per pair of steps, 4 bits a cell instead of 8).
# Tests of the strip kernels, the fuzz slice, the shapes' step times, the far-gap round of the chr1-sized stage at 8 and 16 GiB.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06pk; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_extz2.py tests/test_gpu_fuzz_slice.py -m gpu -x -q -k "strip or fuzz or chain or wide" 2>&1 | grep -E "passed|failed|rror" | head -5
timeout 900 python3 profiles/shapes_bench.py 1000000 3000 100000 2>&1 | cut -c1-200
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
    echo "$label: $(grep -o 'Finished BED.*' $out/$label.$i.txt | grep -o 'in [0-9.]*s') $(grep -o 'process:.*' $out/$label.$i.txt) | $(grep 'n=10813' $out/$label.$i.txt | grep -o 'h2d+device.*') sha $(sha256sum < $d/out_$label.bed | cut -c1-12)"
  done
}
run ws8 SDF_X=0
run ws16 SDF_STAGE_WS_GIB=16
run ws6 SDF_STAGE_WS_GIB=6
