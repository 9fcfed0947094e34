# usage: bash profiles/stage_lanes.sh  -- stage timing for several lane counts
cd $GRAFT_REPO_ROOT
for l in 2 3 4; do
  echo "== SDF_LANES=$l"
  SDF_LANES=$l python3 profiles/stage_bench.py 300000000 40000 3 2>&1 | grep -A1 "run [12]" | grep Finished
done
