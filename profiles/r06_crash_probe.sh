cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
d=/tmp/sdf_stage_one
python3 - > /dev/null 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
d = "/tmp/sdf_stage_one"; os.makedirs(d + "/one", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa)
PY
SDF_LANES=3 SDF_SUPER_BATCH=256 SDF_DEBUG_TIMING=1 sedef_amd/bin/sedef align generate -k 11 $d/genome.fa $d/one/bucket_0000 > /tmp/o.bed 2> gpurun_out/crash.log
tr '\r' '\n' < gpurun_out/crash.log | grep -v "Processing\|DevBuf" | tail -60 | cut -c1-240
