# GPU fuzz campaign of round 5: the banded stripe kernel's rebuilt rows (per-block scalar table, packed x / v shift, the
# upper-edge flavour) forced on EVERY banded task at every stripe width, the default routing next to it, the full-band
# kernels (lane plan by counting sort), the quad kernel forced, and the whole stage on one and three lanes.
#   gpurun --timeout 2400 -- 'bash profiles/r05_fuzz_campaign.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05fuzz
o=gpurun_out/r05fuzz
(SEED=51 ROUNDS=${R:-120} SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 SDF_BSTRIPE_NREG=1 timeout 2100 python tests/fuzz/fuzz_banded.py > $o/bstripe_nreg1.log 2>&1 &
 SEED=52 ROUNDS=${R:-120} SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 SDF_BSTRIPE_NREG=2 timeout 2100 python tests/fuzz/fuzz_banded.py > $o/bstripe_nreg2.log 2>&1 &
 SEED=53 ROUNDS=${R:-120} SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 SDF_BSTRIPE_NREG=4 timeout 2100 python tests/fuzz/fuzz_banded.py > $o/bstripe_nreg4.log 2>&1 &
 SEED=54 ROUNDS=${R:-120} SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 timeout 2100 python tests/fuzz/fuzz_mixed.py > $o/bstripe_mixed.log 2>&1 &
 SEED=55 ROUNDS=${R:-120} timeout 2100 python tests/fuzz/fuzz_banded.py > $o/banded_default.log 2>&1 &
 SEED=56 ROUNDS=${R:-120} SDF_NO_QUAD=0 timeout 2100 python tests/fuzz/fuzz_mixedpair.py > $o/mixedpair_quad.log 2>&1 &
 SEED=57 ROUNDS=${R2:-60} timeout 2100 python tests/fuzz/fuzz_full_band.py > $o/full_band.log 2>&1 &
 SEED=58 ROUNDS=${R3:-40} timeout 2100 python tests/fuzz/fuzz_stage.py > $o/stage.log 2>&1 &
 wait)
for f in $o/*.log; do echo "$f: $(tail -n 1 $f)"; done
