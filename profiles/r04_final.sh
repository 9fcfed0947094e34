# Round 4, the closing check on a fresh box: the driver's own commands, then the shape probes.   bash profiles/r04_final.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04final; mkdir -p $out
timeout 2400 python -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json; cut -c1-260 $out/bench.json
python bench.py --workload hg19mix --tasks 1000000 --steps 10 --warmup 3 --no-pcie-pass > $out/hg19_bench.log 2>&1; tail -1 $out/hg19_bench.log > $out/hg19_bench.json; cut -c1-200 $out/hg19_bench.json
python3 profiles/mix_probe.py mm8 100000 > $out/mm8_100k.log 2>&1; tail -1 $out/mm8_100k.log
python3 profiles/mix_probe.py mm8 100000 64 > $out/mm8_100k_64.log 2>&1; tail -1 $out/mm8_100k_64.log
python3 profiles/mix_probe.py mm8 3000 > $out/mm8_3k.log 2>&1; tail -1 $out/mm8_3k.log
python3 profiles/mix_probe.py hg19 1000000 > $out/hg19.log 2>&1; tail -1 $out/hg19.log
python3 profiles/stage_bench.py --chr1 --one-bucket 5 2>&1 | grep "^run\|chr1-sized" > $out/stage_chr1.txt; cat $out/stage_chr1.txt | cut -c1-200
python3 profiles/stage_bench.py 100000000 40000 6 2>&1 | grep "^run\|^genome" > $out/stage_40k.txt; cat $out/stage_40k.txt | cut -c1-200
