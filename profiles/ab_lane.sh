#!/bin/bash
# A/B of library variants on the lane kernel's shapes (profiles/lane_probe.py), alternating on ONE box: bash profiles/ab_lane.sh <name> <name> ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp sedef_amd/lib/libsedef_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp sedef_amd/lib/ab/$v.so sedef_amd/lib/libsedef_hip.so
    echo "== $v"; python3 profiles/lane_probe.py 400000 2>/dev/null | grep "small\|tiny" | cut -c1-110
  done
done
cp /tmp/keep.so sedef_amd/lib/libsedef_hip.so
