cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04x; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_extz2.py -x -q -m gpu -k "strip or config4 or start_paths" > $out/tests.log 2>&1; tail -3 $out/tests.log
