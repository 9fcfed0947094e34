#!/bin/bash
# Round 6 (end) fuzz after the chained strips' packed flags and the tighter pairing / flag bound: full-band tasks forced onto the
# strip kernels at four columns a lane, eight, and by the chunk; the stage on random genomes; the pytest slice on another seed.
#   gpurun --timeout 1500 -- 'bash profiles/r06_fuzz_final.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06fuzz2; mkdir -p $o
(SEED=71 ROUNDS=${R:-24} timeout 1300 python3 tests/fuzz/fuzz_stage.py > $o/stage.log 2>&1 &
 SEED=72 ROUNDS=${R2:-60} MAXLEN=3500 SDF_STRIP_ALWAYS=1 SDF_STRIP_COLS=4 timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/strips_cols4.log 2>&1 &
 SEED=73 ROUNDS=${R2:-60} MAXLEN=3500 SDF_STRIP_ALWAYS=1 SDF_STRIP_COLS=8 timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/strips_cols8.log 2>&1 &
 SEED=74 ROUNDS=${R2:-60} MAXLEN=2500 SDF_STRIP_ALWAYS=1 timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/strips_auto.log 2>&1 &
 SEED=75 ROUNDS=${R2:-60} timeout 1300 python3 tests/fuzz/fuzz_full_band.py > $o/full_band.log 2>&1 &
 wait)
SDF_FUZZ_SLICE_SEED=707 timeout 600 python3 -m pytest tests/test_gpu_fuzz_slice.py -m gpu -x -q -s 2>&1 | grep -E "fuzz slice|passed|failed" > $o/slice.log
for f in $o/*.log; do echo "$f: $(tail -n 1 $f)"; done
