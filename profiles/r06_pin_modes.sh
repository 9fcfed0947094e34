#!/bin/bash
# Round 6: where the stage's pinned staging comes from (sdf_config.pin_register): hipHostMalloc, registered huge pages for the
# small staging only, or for everything -- alternating on one box: stage clock, set-up time, the anchors call.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
d=/tmp/sdf_stage_one
python3 - > /dev/null 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
d = "/tmp/sdf_stage_one"; os.makedirs(d + "/one", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa)
PY
for rep in 1 2 3 4 5; do
  for m in 0 1 2; do
    SDF_PIN_REGISTER=$m SDF_DEBUG_TIMING=1 sedef_amd/bin/sedef align generate -k 11 $d/genome.fa $d/one/bucket_0000 2>/tmp/err.log >/dev/null
    tr '\r' '\n' < /tmp/err.log > /tmp/err.txt
    echo "pin=$m: $(grep -o 'process:.*' /tmp/err.txt) $(grep -o 'sdf_reserve tasks.*' /tmp/err.txt | grep -o 'rc 0, [0-9.]* ms') | fetched $(grep 'sequences fetched' /tmp/err.txt | grep -o '[0-9.]* ms') anchors $(grep 'anchors done' /tmp/err.txt | grep -o '[0-9.]* ms') $(grep -o 'upload enqueued in [0-9.]* ms' /tmp/err.txt) $(grep -o 'compaction + anchors to the host [0-9.]* ms' /tmp/err.txt | tail -1) dp1 $(grep -m1 'DP round done' /tmp/err.txt | grep -o '[0-9.]* ms')"
  done
done
