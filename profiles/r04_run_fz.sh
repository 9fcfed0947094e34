cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04fz; mkdir -p $out
# full-band tasks up to 14 kb a side: targets beyond 8192 on the chained strips (round 4), blocks of 256 columns by default
(SEED=41 ROUNDS=14 MAXLEN=14000 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/full_14k_a.log 2>&1; tail -2 $out/full_14k_a.log) &
(SEED=42 ROUNDS=14 MAXLEN=14000 SDF_STRIP_COLS=8 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/full_14k_b.log 2>&1; tail -2 $out/full_14k_b.log) &
(SEED=43 ROUNDS=40 MAXLEN=3000 SDF_STRIP_ALWAYS=1 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/full_3k_strips.log 2>&1; tail -2 $out/full_3k_strips.log) &
(SEED=44 ROUNDS=10 MAXLEN=20000 SDF_STRIPE_SPIN_CAP=64 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/full_20k_giveup.log 2>&1; tail -2 $out/full_20k_giveup.log) &
wait
