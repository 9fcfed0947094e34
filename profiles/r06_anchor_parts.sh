#!/bin/bash
# Round 6 (end): the seed anchors of a super-batch in 1 / 2 / 4 / 8 parts (SDF_ANCHOR_PARTS), each under the chaining of the one before:
# chr1-sized bucket, stage clock and timeline; the stage tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06parts; mkdir -p $out
timeout 900 python3 -m pytest tests/test_stage_pairs.py tests/test_stage_scale.py tests/test_dropin.py tests/test_host_pipeline.py tests/test_pinning.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -3
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
for rep in 1 2; do
for parts in 1 2 4 8 0; do
  for i in 1 2 3; do
    SDF_ANCHOR_PARTS=$parts SDF_DEBUG_TIMING=1 $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_p$parts.bed 2> $out/p$parts.$i.log
    tr '\r' '\n' < $out/p$parts.$i.log | grep -v "Processing\|DevBuf" > $out/p$parts.$i.txt; rm $out/p$parts.$i.log
    echo "parts=$parts: $(grep -o 'process:.*' $out/p$parts.$i.txt | grep -o 'stage [0-9.]*s') | $(grep -o 'stage *[0-9.]* ms. batch@0 anchors done, parts chained' $out/p$parts.$i.txt | grep -o '[0-9.]* ms') anchors+chained | sha $(sha256sum < $d/out_p$parts.bed | cut -c1-12)"
  done
done
done
