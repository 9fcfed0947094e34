cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04l; mkdir -p $out
for ws in 64 96 128 200; do
python3 profiles/mix_probe.py mm8 100000 $ws > $out/a_$ws.log 2>&1; echo "ws $ws: $(tail -1 $out/a_$ws.log)"
done
GPU_MAX_HW_QUEUES=8 python3 profiles/mix_probe.py mm8 100000 128 > $out/b8.log 2>&1; echo "hq8 ws 128: $(tail -1 $out/b8.log)"
python3 profiles/mix_probe.py mm8 3000 > $out/m3k.log 2>&1; tail -1 $out/m3k.log
python3 profiles/mix_probe.py hg19 1000000 > $out/hg.log 2>&1; tail -1 $out/hg.log
python3 bench.py --no-cpu-baseline --no-pcie-pass > $out/bench.log 2>&1; tail -1 $out/bench.log | cut -c1-150
