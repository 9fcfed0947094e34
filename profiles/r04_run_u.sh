cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04v; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_extz2.py -x -q -m gpu -k "strip" > $out/strip_tests.log 2>&1; tail -5 $out/strip_tests.log
SDF_DEBUG_CLASSES=1 SDF_DEBUG_TIMING=1 python3 profiles/stage_bench.py --chr1 --one-bucket 3 > $out/chr1_wide.log 2>&1
grep "Finished BED\|sdf_extz2_batch n=10813" $out/chr1_wide.log
