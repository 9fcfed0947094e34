cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04x; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_host_pipeline.py tests/test_stage_scale.py tests/test_pinning.py -x -q -m gpu > $out/stage_tests.log 2>&1; tail -3 $out/stage_tests.log
SDF_DEBUG_TIMING=1 python3 profiles/stage_bench.py --chr1 --one-bucket 5 > $out/chr1.log 2>&1
grep "Finished BED" $out/chr1.log
python3 profiles/stage_bench.py 100000000 40000 6 2>&1 | grep "Finished BED"
