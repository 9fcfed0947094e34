cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r04u; mkdir -p $out
for i in 1 2 3 4; do python3 profiles/mix_probe.py mm8 3000 > $out/m3k_$i.log 2>&1; tail -1 $out/m3k_$i.log; done
for i in 1 2 3; do python3 profiles/mix_probe.py hg19 1000000 > $out/hg_$i.log 2>&1; tail -1 $out/hg_$i.log; done
