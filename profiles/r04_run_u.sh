cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04x; mkdir -p $out
for f in 1 2 1 2; do echo -n "inflight $f: "; python bench.py --inflight $f --no-pcie-pass --no-cpu-baseline 2>&1 | tail -1 | cut -c1-190; done
for f in 1 2; do echo -n "hg19 inflight $f: "; python bench.py --inflight $f --workload hg19mix --tasks 1000000 --steps 10 --warmup 3 --no-pcie-pass --no-cpu-baseline 2>&1 | tail -1 | cut -c1-190; done
timeout 1500 python3 -m pytest tests/test_bench_launch.py -x -q -m gpu > $out/bench_tests.log 2>&1; tail -3 $out/bench_tests.log
