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
for ws in 16 8 16 8; do
  SDF_STAGE_WS_GIB=$ws SDF_DEBUG_TIMING=1 sedef_amd/bin/sedef align generate -k 11 $d/genome.fa $d/one/bucket_0000 2>&1 >/dev/null | tr '\r' '\n' | grep "DevBuf\|anchors_range\|anchors:\|n=1860\|anchors done\|fetched\|process:" | grep -v "0.0[0-9] ms\]" | sed "s/^/ws$ws /" | cut -c1-150
done
