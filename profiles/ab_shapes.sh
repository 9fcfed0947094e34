#!/bin/bash
# A/B of library variants (profiles/ab_build.sh) on the shapes next to the headline, alternating on ONE box:
#   bash profiles/ab_shapes.sh <name> <name> ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp sedef_amd/lib/libsedef_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp sedef_amd/lib/ab/$v.so sedef_amd/lib/libsedef_hip.so
    echo "== $v"; python3 profiles/shapes_bench.py 1000000 3000 2>/dev/null | cut -c1-150
    python3 profiles/mix_probe.py mm8 100000 2>/dev/null | tail -1 | cut -c1-150
  done
done
cp /tmp/keep.so sedef_amd/lib/libsedef_hip.so
