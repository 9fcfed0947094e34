#!/bin/bash
# Round 6 (end) fuzz after the score tables of the wave / stripe / banded-stripe kernels: each forced in turn.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06fuzz3; mkdir -p $o
(SEED=81 ROUNDS=${R:-40} SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 timeout 1300 python3 tests/fuzz/fuzz_banded.py > $o/bstripe_all.log 2>&1 &
 SEED=82 ROUNDS=${R:-40} SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 SDF_BSTRIPE_NREG=2 timeout 1300 python3 tests/fuzz/fuzz_banded.py > $o/bstripe_nreg2.log 2>&1 &
 SEED=83 ROUNDS=${R:-40} SDF_NO_STRIP=1 SDF_STRIPE_MIN=128 timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/stripe.log 2>&1 &
 SEED=84 ROUNDS=${R:-40} SDF_NO_STRIP=1 SDF_STRIPE_MIN=128 SDF_STRIPE_NREG=1 timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/stripe_nreg1.log 2>&1 &
 SEED=85 ROUNDS=${R:-40} SDF_NO_PAIR=1 timeout 1300 python3 tests/fuzz/fuzz_banded.py > $o/wave.log 2>&1 &
 SEED=86 ROUNDS=${R:-40} timeout 1300 python3 tests/fuzz/fuzz_mixed.py > $o/mixed.log 2>&1 &
 wait)
for f in $o/*.log; do echo "$f: $(tail -n 1 $f)"; done
