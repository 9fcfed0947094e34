# usage: bash profiles/stage_threads.sh  -- stage-level timing of `sedef align generate` for several host thread counts
cd $GRAFT_REPO_ROOT
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for t in 16 32 64; do
  echo "== SDF_HOST_THREADS=$t"
  SDF_HOST_THREADS=$t python3 profiles/stage_bench.py 100000000 10000 2 2>&1 | grep -A3 "run 1"
done
