cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04x; mkdir -p $out
for s in 0 1 0 1; do echo "serial uploads $s:"; SDF_SERIAL_UPLOADS=$s python3 profiles/stage_bench.py 100000000 40000 4 2>&1 | grep "Finished BED" | sed 's/.* in //' | cut -c1-6 | tr '\n' ' '; echo; done
