cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(SEED=31 ROUNDS=40 SDF_STRIP_ALWAYS=1 timeout 900 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_strip8.log 2>&1 &
 SEED=32 ROUNDS=40 SDF_STRIP_ALWAYS=1 SDF_STRIP_COLS=4 timeout 900 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_strip4.log 2>&1 &
 SEED=33 ROUNDS=60 MAXLEN=250 SDF_LANE_MIN=1 timeout 900 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_lane.log 2>&1 &
 SEED=34 ROUNDS=40 MAXLEN=6000 SDF_STRIP_ALWAYS=1 SDF_STRIPE_SPIN_CAP=2000 timeout 900 python tests/fuzz/fuzz_full_band.py > gpurun_out/fuzz_chain_giveup.log 2>&1 &
 SEED=35 ROUNDS=40 timeout 900 python tests/fuzz/fuzz_mixed.py > gpurun_out/fuzz_mixed.log 2>&1 &
 SEED=36 ROUNDS=40 timeout 900 python tests/fuzz/fuzz_banded.py > gpurun_out/fuzz_banded.log 2>&1 &
 wait)
tail -2 gpurun_out/fuzz_*.log
