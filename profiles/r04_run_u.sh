cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04w; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_pinning.py tests/test_host_pipeline.py tests/test_stage_scale.py -x -q -m gpu > $out/stage_tests.log 2>&1; tail -3 $out/stage_tests.log
SDF_DEBUG_TIMING=1 python3 profiles/stage_bench.py --chr1 --one-bucket 4 > $out/chr1.log 2>&1
grep "Finished BED\|sdf_anchors_batch n=1860" $out/chr1.log
python3 profiles/stage_bench.py 100000000 40000 6 > $out/s40k.log 2>&1
grep "Finished BED" $out/s40k.log
