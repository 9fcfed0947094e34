# usage: bash profiles/stage_lanes.sh [genome_len] [n_dup]  -- stage timing for several lane counts
cd $GRAFT_REPO_ROOT
G=${1:-300000000}; N=${2:-40000}
for l in 1 2 3 4; do
  echo "== SDF_LANES=$l"
  SDF_LANES=$l python3 profiles/stage_bench.py $G $N 4 2>&1 | grep Finished | tail -3 | cut -c1-120
done
