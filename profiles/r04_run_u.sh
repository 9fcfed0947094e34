cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04x; mkdir -p $out
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
