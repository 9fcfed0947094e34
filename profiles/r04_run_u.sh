cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r04x; mkdir -p $out
SDF_DEBUG_PLAN=1 SDF_DEBUG_CLASSES=1 python3 profiles/mix_probe.py hg19 500000 > $out/hg500_plan.log 2>&1
SDF_DEBUG_PLAN=1 SDF_DEBUG_CLASSES=1 python3 profiles/mix_probe.py hg19 330000 > $out/hg330_plan.log 2>&1
