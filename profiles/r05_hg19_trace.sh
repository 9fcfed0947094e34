#!/bin/bash
# Round 5: kernel timeline of ONE call of the hg19-shaped mixture (1,000,000 tasks), lane stream on the highest priority (default)
# and as in round 4 (SDF_LANE_PRIO=0).   bash profiles/r05_hg19_trace.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
out=gpurun_out/r05t; mkdir -p $out
M="python3 profiles/mix_probe.py hg19 1000000"
for tag in bins sort; do
  if [ $tag = sort ]; then export SDF_LANE_PLAN=sort; else unset SDF_LANE_PLAN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$tag -o run -- $M > $out/st_$tag.log 2>&1
  cp $(find $out/st_$tag -name "*kernel_stats.csv" | head -1) $out/hg19_kernel_stats_$tag.csv
  python3 profiles/timeline.py $out/st_$tag 140 | grep -v lt_config > $out/timeline_$tag.txt 2>&1
  rm -rf $out/st_$tag
  tail -1 $out/st_$tag.log
done
