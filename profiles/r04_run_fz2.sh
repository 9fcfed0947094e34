cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04fz2; mkdir -p $out
# the builds of the end of round 4 (chains to 65,536 bases / 256-column blocks, the banded stripes' and the wave kernel's priorities, the threaded scan) through
# the general campaigns once more, four side by side (the box has 16 CPUs: the oracle is the slow side)
(SEED=61 ROUNDS=500 N=500 timeout 1700 python3 tests/fuzz/fuzz_mixed.py > $out/mixed.log 2>&1; tail -1 $out/mixed.log) &
(SEED=62 ROUNDS=500 timeout 1700 python3 tests/fuzz/fuzz_banded.py > $out/banded.log 2>&1; tail -1 $out/banded.log) &
(SEED=63 ROUNDS=600 MAXLEN=2600 timeout 1700 python3 tests/fuzz/fuzz_full_band.py > $out/full.log 2>&1; tail -1 $out/full.log) &
(SEED=64 ROUNDS=300 SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 timeout 1700 python3 tests/fuzz/fuzz_banded.py > $out/bstripe_all.log 2>&1; tail -1 $out/bstripe_all.log) &
(SEED=65 ROUNDS=400 SDF_MIXED_MIN=2 timeout 1700 python3 tests/fuzz/fuzz_mixedpair.py > $out/mixedpair.log 2>&1; tail -1 $out/mixedpair.log) &
wait
