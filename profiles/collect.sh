#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>
# kernel-trace stats of the isolated-launch mode (SDF_PIPELINE=0: what bench.py's roofline block times) and of
# the default pipelined mode, the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ set) in runs of their own, and kernel-trace
# stats of the hg19-shaped mixture (configs[3]) and the mm8-like mixed-band batch (configs[4]).
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # (setenv inside the program is too late under rocprofv3: its tool library starts HIP first)
out=gpurun_out/$tag
mkdir -p $out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-pass --no-stage"
BI="$B --inflight 1"   # (the isolated-launch passes: one call at a time, or two whole-batch launches would overlap)
B1="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-pass --no-stage --inflight 1"
SDF_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_iso -o run -- $BI > $out/stats_iso.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_pipe -o run -- $B > $out/stats_pipe.log 2>&1
SDF_PIPELINE=0 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/fetch -o run -- $B1 > $out/fetch.log 2>&1
SDF_PIPELINE=0 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/write -o run -- $B1 > $out/write.log 2>&1
SDF_PIPELINE=0 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $out/sq -o run -- $B1 > $out/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_hg19 -o run -- python3 profiles/mix_probe.py hg19 1000000 > $out/stats_hg19.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mm8 -o run -- python3 profiles/mix_probe.py mm8 3000 > $out/stats_mm8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_mm8big -o run -- python3 profiles/mix_probe.py mm8 100000 > $out/stats_mm8big.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_fullband -o run -- python3 profiles/chain_probe.py > $out/stats_fullband.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_lane -o run -- python3 profiles/lane_probe.py 400000 > $out/stats_lane.log 2>&1
for d in stats_iso stats_pipe stats_hg19 stats_mm8 stats_mm8big stats_fullband stats_lane; do cp $(find $out/$d -name "*kernel_stats.csv" | head -1) $out/${d}_kernel_stats.csv; done
python3 profiles/pmc_summary.py $out/fetch > $out/pmc_fetch.txt
python3 profiles/pmc_summary.py $out/write > $out/pmc_write.txt
python3 profiles/pmc_summary.py $out/sq extz2_ 199900000 > $out/pmc_sq.txt
python3 profiles/timeline.py $out/stats_pipe 14 > $out/timeline_pipe.txt
python3 profiles/timeline.py $out/stats_hg19 40 > $out/timeline_hg19.txt
python3 profiles/timeline.py $out/stats_mm8 24 > $out/timeline_mm8.txt
grep -h "^{\"metric" $out/stats_iso.log > $out/bench_iso.json
grep -h "^{\"metric" $out/stats_pipe.log > $out/bench_pipe.json
grep -h "tasks" $out/stats_hg19.log $out/stats_mm8.log $out/stats_mm8big.log $out/stats_fullband.log $out/stats_lane.log > $out/mix_lines.txt
python3 profiles/make_traffic_json.py $out profiles/${tag}_pair_kernel_pmc.txt > $out/hbm_traffic_line.txt
cat $out/stats_iso_kernel_stats.csv | head -8; cat $out/pmc_fetch.txt $out/pmc_write.txt $out/pmc_sq.txt | grep "pair_kernel\|traceback"; cat $out/mix_lines.txt
cp profiles/hbm_traffic.json $out/hbm_traffic.json
rm -rf $out/stats_iso $out/stats_pipe $out/fetch $out/write $out/sq $out/stats_hg19 $out/stats_mm8 $out/stats_mm8big $out/stats_fullband $out/stats_lane
