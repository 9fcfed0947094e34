#!/bin/bash
# A/B of library variants built by profiles/ab_build.sh on ONE box, alternating: bash profiles/ab_bench.sh <name> <name> ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp sedef_amd/lib/libsedef_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp sedef_amd/lib/ab/$v.so sedef_amd/lib/libsedef_hip.so
    python3 bench.py --no-stage --no-cpu-baseline --no-pcie-pass --steps 10 --warmup 3 ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['value_one_call_in_flight'], d['roofline']['avg_launch_ms'], d['spot_check']['ok'])"
  done
done
cp /tmp/keep.so sedef_amd/lib/libsedef_hip.so
