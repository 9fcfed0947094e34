cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04fz3; mkdir -p $out
# the stripe kernel on targets of 8,193 .. 32,512 bases (few wide full-band tasks per batch), every stripe width; the few-long-tasks rule inside chunks of chains
(SEED=71 ROUNDS=10 MAXLEN=14000 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/wide_default.log 2>&1; tail -1 $out/wide_default.log) &
(SEED=72 ROUNDS=10 MAXLEN=14000 SDF_STRIPE_NREG=1 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/wide_nreg1.log 2>&1; tail -1 $out/wide_nreg1.log) &
(SEED=73 ROUNDS=10 MAXLEN=14000 SDF_STRIPE_NREG=4 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/wide_nreg4.log 2>&1; tail -1 $out/wide_nreg4.log) &
(SEED=74 ROUNDS=8 MAXLEN=24000 SDF_CHAIN_MIN=16 timeout 1500 python3 tests/fuzz/fuzz_full_band.py > $out/wide_mixed_routes.log 2>&1; tail -1 $out/wide_mixed_routes.log) &
wait
