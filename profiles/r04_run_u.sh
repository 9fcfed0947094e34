cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for l in 1 2 1 2; do echo -n "lanes $l: "; SDF_LANES=$l python3 profiles/stage_bench.py --chr1 --one-bucket 4 2>&1 | grep "Finished BED" | sed 's/.* in //' | cut -c1-6 | tr '\n' ' '; echo; done
