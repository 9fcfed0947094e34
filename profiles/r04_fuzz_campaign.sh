# GPU fuzz campaign of round 4: the mixed pairs (extz2_pair.hip MIXED: two banded tasks of one band and different lengths
# per wavefront) forced from two candidates per chunk (SDF_MIXED_MIN=2), every script of tests/fuzz that draws banded
# tasks, side by side; plus the chained strips with a short spin cap (stripes give up, tasks run again: ADVICE r3's race).
#   gpurun --timeout 2400 -- 'bash profiles/r04_fuzz_campaign.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04fuzz
o=gpurun_out/r04fuzz
(SEED=41 ROUNDS=${R:-170} N=600 LMAX=1800 timeout 2000 python tests/fuzz/fuzz_mixedpair.py > $o/mixedpair_a.log 2>&1 &
 SEED=42 ROUNDS=${R:-170} N=600 LMAX=900 timeout 2000 python tests/fuzz/fuzz_mixedpair.py > $o/mixedpair_b.log 2>&1 &
 SEED=43 ROUNDS=${R2:-40} N=300 LMAX=7000 timeout 2000 python tests/fuzz/fuzz_mixedpair.py > $o/mixedpair_long.log 2>&1 &
 SEED=44 ROUNDS=${R:-170} SDF_MIXED_MIN=2 timeout 2000 python tests/fuzz/fuzz_banded.py > $o/banded.log 2>&1 &
 SEED=45 ROUNDS=${R:-170} SDF_MIXED_MIN=2 timeout 2000 python tests/fuzz/fuzz_mixed.py > $o/mixed.log 2>&1 &
 SEED=46 ROUNDS=${R2:-40} MAXLEN=6000 SDF_STRIP_ALWAYS=1 SDF_STRIPE_SPIN_CAP=2000 timeout 2000 python tests/fuzz/fuzz_full_band.py > $o/chain_giveup.log 2>&1 &
 wait)
for f in $o/*.log; do echo "$f: $(tail -n 1 $f)"; done
