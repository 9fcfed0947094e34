# GPU fuzz campaign of the round: the three scripts of tests/fuzz with the round's kernels forced (strips with 8 and 4 columns,
# lane kernel from one task, chained strips with a short spin cap so that stripes give up and tasks are run again), side by side.
#   gpurun --timeout 1800 -- 'bash profiles/fuzz_campaign.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(SEED=31 ROUNDS=${R40:-320} SDF_STRIP_ALWAYS=1 timeout 1500 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_strip8.log 2>&1 &
 SEED=32 ROUNDS=${R40:-320} SDF_STRIP_ALWAYS=1 SDF_STRIP_COLS=4 timeout 1500 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_strip4.log 2>&1 &
 SEED=33 ROUNDS=${R60:-480} MAXLEN=250 SDF_LANE_MIN=1 timeout 1500 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_lane.log 2>&1 &
 SEED=34 ROUNDS=${R40:-320} MAXLEN=6000 SDF_STRIP_ALWAYS=1 SDF_STRIPE_SPIN_CAP=2000 timeout 1500 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_chain_giveup.log 2>&1 &
 SEED=35 ROUNDS=${R40:-320} timeout 1500 python tests/fuzz/fuzz_mixed.py > gpurun_out/fuzz_mixed.log 2>&1 &
 SEED=36 ROUNDS=${R40:-320} timeout 1500 python tests/fuzz/fuzz_banded.py > gpurun_out/fuzz_banded.log 2>&1 &
 wait)
for f in gpurun_out/fuzz_*.log; do tail -n 1 $f; done
