# usage: bash profiles/shape_probe.sh [qlen ...]   -- SQ instruction counters of the DP kernels per anti-diagonal
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for q in ${@:-1000}; do
  n=$((20000000/q/2))
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/sq_$q -o run -- python3 bench.py --steps 1 --warmup 0 --tasks $n --qlen $q --no-cpu-baseline > gpurun_out/sq_$q.log 2>&1
  echo "== q=$q n=$n rows=$((n*(2*q-1)))"
  python3 profiles/pmc_summary.py gpurun_out/sq_$q extz2_ $((n*(2*q-1)))
done
