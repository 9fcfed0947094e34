#!/bin/bash
# Round 4, late: the hg19 mixture after the chains went to blocks of 256 columns: SQ / FETCH / WRITE counters per kernel of the
# 1,000,000-task call, and the stage-level CPU leg of the same day.   bash profiles/r04_collect_c.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r04c; mkdir -p $out
M="python3 profiles/mix_probe.py hg19 1000000"
pmc() { d=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "${ctr[@]}" -d $out/$d -o run -- "$@" > $out/$d.log 2>&1
  python3 profiles/pmc_summary.py $out/$d sdf:: > $out/pmc_$d.txt; rm -rf $out/$d; }
pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- $M
pmc fetch FETCH_SIZE -- $M
pmc write WRITE_SIZE -- $M
cat $out/pmc_sq.txt $out/pmc_fetch.txt $out/pmc_write.txt > $out/hg19_pmc_cols4.txt
python3 profiles/stage_bench.py --chr1 --one-bucket --cpu 2 > $out/stage_cpu.log 2>&1; grep "CPU leg\|^run" $out/stage_cpu.log
python3 profiles/stage_bench.py --chr1 --one-bucket 3 > $out/stage_gpu.log 2>&1; grep "^run" $out/stage_gpu.log
