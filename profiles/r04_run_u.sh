cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r04x; mkdir -p $out
SDF_DEBUG_PLAN=1 python3 profiles/mix_probe.py hg19 1000000 > $out/hg1m_plan.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/hgprof -o run -- python3 profiles/mix_probe.py hg19 1000000 > $out/hgprof.log 2>&1
cp $(find $out/hgprof -name "*kernel_stats.csv" | head -1) $out/hg19_kernel_stats.csv; rm -rf $out/hgprof
head -8 $out/hg19_kernel_stats.csv | cut -c1-200
