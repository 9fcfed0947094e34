#!/bin/bash
# Round 4, first GPU pass (run through gpurun from the repo root): the counters VERDICT r3 asked for BEFORE any kernel work.
#   bash profiles/r04_collect_a.sh
#  1. the SQ stall counters of the isolated headline launch (extz2_pair_kernel<3>): is the SIMD issue-saturated?
#  2. the mm8-like batch of 100,000 tasks: tasks / rows / cells per launch class (SDF_DEBUG_CLASSES) and, per kernel,
#     VALU / SALU instructions, wave cycles, FETCH_SIZE, WRITE_SIZE (separate --pmc passes)
#  3. the north star's batch as a bench line with its own cpu_baseline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r04a
mkdir -p $out
rocprofv3 -L > $out/counters_all.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TCC_[A-Z0-9_]*" $out/counters_all.txt | sort -u > $out/counter_names.txt
B1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-pass"
pmc() {  # pmc <dir> <counters...> -- <cmd...>
  d=$1; shift
  ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rocprofv3 --kernel-trace --output-format csv --pmc "${ctr[@]}" -d $out/$d -o run -- "$@" > $out/$d.log 2>&1
  python3 profiles/pmc_summary.py $out/$d sdf:: > $out/pmc_$d.txt
  rm -rf $out/$d
}
# ---- 1. headline launch ----
export SDF_PIPELINE=0
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES -- $B1
pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -- $B1
pmc sq3 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -- $B1
pmc sq4 SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $B1
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT -- $B1
unset SDF_PIPELINE
# ---- 2. mm8-like batch, 100,000 tasks ----
M="python3 profiles/mix_probe.py mm8 100000"
SDF_DEBUG_CLASSES=1 $M > $out/mm8_classes.log 2> $out/mm8_classes.err
grep "^\[class" $out/mm8_classes.err > $out/mm8_classes.txt
pmc mm8_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- $M
pmc mm8_fetch FETCH_SIZE -- $M
pmc mm8_write WRITE_SIZE -- $M
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mm8big -o run -- $M > $out/stats_mm8big.log 2>&1
cp $(find $out/stats_mm8big -name "*kernel_stats.csv" | head -1) $out/mm8_100k_kernel_stats.csv
rm -rf $out/stats_mm8big
M3="python3 profiles/mix_probe.py mm8 3000"
SDF_DEBUG_CLASSES=1 $M3 > $out/mm8_3k_classes.log 2> $out/mm8_3k_classes.err
grep "^\[class" $out/mm8_3k_classes.err > $out/mm8_3k_classes.txt
# ---- 3. hg19 mixture: bench line with its own cpu_baseline; its classes; PMC of its kernels ----
python3 bench.py --workload hg19mix --tasks 1000000 --steps 10 --warmup 3 --no-pcie-pass > $out/hg19_bench.log 2>&1
tail -1 $out/hg19_bench.log > $out/hg19_bench.json
H="python3 profiles/mix_probe.py hg19 1000000"
SDF_DEBUG_CLASSES=1 $H > $out/hg19_classes.log 2> $out/hg19_classes.err
grep "^\[class" $out/hg19_classes.err > $out/hg19_classes.txt
pmc hg19_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- $H
pmc hg19_fetch FETCH_SIZE -- $H
pmc hg19_write WRITE_SIZE -- $H
# ---- 4. the default bench line on this box ----
python3 bench.py > $out/bench_default.log 2>&1
tail -1 $out/bench_default.log > $out/bench_default.json
ls -la $out
cat $out/pmc_sq1.txt $out/pmc_sq3.txt $out/pmc_grbm.txt | grep "pair_kernel<3"
grep -h "tasks" $out/mm8_classes.log $out/mm8_3k_classes.log $out/hg19_classes.log
cat $out/hg19_bench.json | cut -c1-600
