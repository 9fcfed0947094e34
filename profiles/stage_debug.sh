# usage: bash profiles/stage_debug.sh [genome_len] [n_dup]: one run of the CLI with SDF_DEBUG_TIMING=1 (per-call breakdown)
cd $GRAFT_REPO_ROOT
G=${1:-100000000}; N=${2:-40000}
python3 - <<PY
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hostgen
from sedef_amd.host import build_host
build_host()
hostgen.make_genome("/tmp/stage_dbg.fa", seed=11, glen=$G, nsd=$N)
PY
export GPU_MAX_HW_QUEUES=8
SDF_DEBUG_TIMING=1 sedef_amd/bin/sedef align generate -k 11 /tmp/stage_dbg.fa /tmp/stage_dbg.fa.bed > /tmp/stage_dbg.out 2> gpurun_out/stage_dbg.log
tr '\r' '\n' < gpurun_out/stage_dbg.log | grep -v Processing | tail -80
