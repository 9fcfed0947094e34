#!/bin/bash
# Round 6, first call: where the round starts.  GPU tests, the bench line, the chr1-sized stage with its phase timeline.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06start; mkdir -p $out
timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gputest.txt 2>&1; tail -3 $out/gputest.txt
timeout 300 python3 bench.py > $out/bench.json 2> $out/bench.err; cut -c1-300 $out/bench.json
timeout 600 python3 profiles/stage_bench.py --chr1 --one-bucket 3 > $out/stage.txt 2>&1; grep "^run" $out/stage.txt
d=$(ls -d /tmp/sdf_stage_* | head -1)
SDF_DEBUG_TIMING=1 sedef_amd/bin/sedef align generate -k 11 $d/genome.fa $d/buckets/bucket_0000 > /tmp/o.bed 2> $out/stage_dbg.log
tr '\r' '\n' < $out/stage_dbg.log | grep -v Processing > $out/stage_dbg.txt; rm $out/stage_dbg.log
wc -l /tmp/o.bed
