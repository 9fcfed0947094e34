#!/bin/bash
# Round 4, the mm8-like batch of 100,000 tasks after the mixed pairs: kernel trace, SQ / FETCH / WRITE counters per kernel,
# tasks / rows / cells per launch class, at a 128 GiB workspace.   bash profiles/r04_collect_b.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=${HQ:-24}
out=gpurun_out/r04b2; mkdir -p $out
M="python3 profiles/mix_probe.py mm8 100000 128"
for i in 1 2 3; do $M > $out/run$i.log 2>&1; tail -1 $out/run$i.log; done
SDF_NO_MIXED=1 $M > $out/nomixed.log 2>&1; tail -1 $out/nomixed.log
python3 profiles/mix_probe.py mm8 3000 > $out/m3k.log 2>&1; tail -1 $out/m3k.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o run -- $M > $out/st.log 2>&1
cp $(find $out/st -name "*kernel_stats.csv" | head -1) $out/mm8_100k_kernel_stats.csv
python3 profiles/timeline.py $out/st 100 > $out/timeline.txt 2>&1
rm -rf $out/st
pmc() { d=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "${ctr[@]}" -d $out/$d -o run -- "$@" > $out/$d.log 2>&1
  python3 profiles/pmc_summary.py $out/$d sdf:: > $out/pmc_$d.txt; rm -rf $out/$d; }
pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- $M
pmc fetch FETCH_SIZE -- $M
pmc write WRITE_SIZE -- $M
SDF_DEBUG_CLASSES=1 $M > $out/classes.log 2> $out/classes.err
grep "^\[class" $out/classes.err > $out/classes.txt
