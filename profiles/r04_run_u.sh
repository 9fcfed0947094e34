cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04w; mkdir -p $out
SDF_DEBUG_TIMING=1 python3 profiles/stage_bench.py 100000000 40000 5 > $out/s40k.log 2>&1
grep "Finished BED" $out/s40k.log
