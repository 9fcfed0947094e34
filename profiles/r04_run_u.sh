cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04w; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_extz2.py -x -q -m gpu -k "brief or strip_chain" > $out/brief_tests.log 2>&1; tail -15 $out/brief_tests.log
