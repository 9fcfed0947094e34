cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SDF_DEBUG_TIMING=1 python3 profiles/stage_bench.py 100000000 40000 4 2>&1 | grep "process:\|^run\|Finished BED\|sdf_create device\|sdf_reserve" | cut -c1-200
