#!/bin/bash
# A/B of library variants (profiles/ab_build.sh) on the hg19-shaped mixture and the full-band batch, alternating on ONE box:
#   bash profiles/ab_hg19.sh <name> <name> ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
cp sedef_amd/lib/libsedef_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp sedef_amd/lib/ab/$v.so sedef_amd/lib/libsedef_hip.so
    echo "$v hg19: $(python3 profiles/hg19_steps.py 1000000 10 2>/dev/null | tail -1)   full band 20000 x 1000^2: $(python3 profiles/fullband_probe.py 20000 1000 2>/dev/null | tail -1 | grep -o '[0-9.]* ms/step')"
  done
done
cp /tmp/keep.so sedef_amd/lib/libsedef_hip.so
