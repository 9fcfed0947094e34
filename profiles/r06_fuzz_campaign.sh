#!/bin/bash
# Round 6 fuzz: the whole stage on the resident-sequence path (random genomes: host pipeline on the reference kernel against the
# GPU provider through the C ABI and the CLI on one and three lanes, with the resident path off as well), and the kernels on
# the default routing / mixed pairs / full band (the planner's bounds and the lane planning changed this round).
#   gpurun --timeout 1500 -- 'bash profiles/r06_fuzz_campaign.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06fuzz; mkdir -p $o
(SEED=61 ROUNDS=${R:-36} timeout 1300 python3 tests/fuzz/fuzz_stage.py > $o/stage.log 2>&1 &
 SEED=62 ROUNDS=${R:-36} SDF_RESIDENT_DP=0 timeout 1300 python3 tests/fuzz/fuzz_stage.py > $o/stage_nores.log 2>&1 &
 SEED=63 ROUNDS=${R:-36} GLEN_MAX=6000000 NSD_MAX=500 timeout 1300 python3 tests/fuzz/fuzz_stage.py > $o/stage_big.log 2>&1 &
 SEED=64 ROUNDS=${R2:-120} timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/full_band.log 2>&1 &
 SEED=65 ROUNDS=${R2:-120} timeout 1300 python3 tests/fuzz/fuzz_banded.py > $o/banded.log 2>&1 &
 SEED=66 ROUNDS=${R2:-120} timeout 1300 python3 tests/fuzz/fuzz_mixedpair.py > $o/mixedpair.log 2>&1 &
 wait)
for f in $o/*.log; do echo "$f: $(tail -n 1 $f)"; done
