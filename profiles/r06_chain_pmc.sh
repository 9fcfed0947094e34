#!/bin/bash
# Round 6 (end): what a wavefront of the chained strips spends its cycles on -- SQ counters of the chain launches of the full-band
# batch (20,000 x 1000 x 1000: eight columns a lane) and of the hg19 mixture (four columns a lane, next to the lanes and strips).
# WAIT_ANY (parked on a wait) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES, in quad-cycles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06chainpmc; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/fb -o run -- python3 $R/profiles/fullband_probe.py 20000 1000 > $out/fb.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/hg -o run -- python3 $R/profiles/mix_probe.py hg19 1000000 > $out/hg.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $out/fb2 -o run -- python3 $R/profiles/fullband_probe.py 20000 1000 > $out/fb2.log 2>&1
cd $R
echo "== full band 20,000 x 1000 x 1000"; tail -1 $out/fb.log | cut -c1-200; python3 profiles/pmc_summary.py $out/fb strip
python3 profiles/pmc_summary.py $out/fb2 strip
echo "== hg19 mixture, 1,000,000 tasks"; tail -1 $out/hg.log | cut -c1-200; python3 profiles/pmc_summary.py $out/hg extz2_
rm -rf $out/fb $out/hg $out/fb2
