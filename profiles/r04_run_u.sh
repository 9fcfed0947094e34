cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04x; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_gpu_extz2.py -x -q -m gpu -k "strip or config4 or start_paths" > $out/tests.log 2>&1; tail -3 $out/tests.log
export GPU_MAX_HW_QUEUES=8
for i in 1 2 3; do for g in 0 1; do echo -n "wg4 $g: "; SDF_CHAIN_WG4=$g python3 profiles/mix_probe.py hg19 1000000 2>&1 | tail -1; done; done
SDF_CHAIN_WG4=1 python3 profiles/placement_probe.py 1000000 2>&1 | tail -5
for g in 0 1; do echo "wg4 $g:"; SDF_CHAIN_WG4=$g python3 profiles/stage_bench.py --chr1 --one-bucket 3 2>&1 | grep "Finished BED" | sed 's/.* in //' | cut -c1-6 | tr '\n' ' '; echo; done
