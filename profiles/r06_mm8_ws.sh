#!/bin/bash
# Round 6: BASELINE configs[4] at 100,000 tasks (131 GB of direction flags) against the workspace budget: default (half of the
# free HBM: three chunks, two side by side) and 65 % of it (every chunk a region of its own: they all start together)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for pct in 50 65; do
  for i in 1 2 3; do echo "clamp=$pct%: $(SDF_WS_CLAMP_PCT=$pct SDF_DEBUG_PLAN=${DBG:-0} python3 profiles/mix_probe.py mm8 100000 200 2>&1 | tail -1)"; done
done
