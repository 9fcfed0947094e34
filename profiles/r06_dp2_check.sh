cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
timeout 1200 python3 -m pytest tests/test_gpu_extz2.py -m gpu -x -q -k "stripe or strip or chain or heavy or config4 or config5 or sedef_shapes or long" 2>&1 | tail -3
bash profiles/r06_stage_dp2.sh 2>&1 | grep "^base\|^ws16:"
