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
for rep in 1 2 3; do for sd in 1 0; do
  HSA_ENABLE_SDMA=$sd SDF_DEBUG_TIMING=1 sedef_amd/bin/sedef align generate -k 11 $d/genome.fa $d/one/bucket_0000 2>/tmp/e.log >/dev/null
  tr '\r' '\n' < /tmp/e.log > /tmp/e.txt
  echo "sdma=$sd: $(grep -o 'process:.*' /tmp/e.txt) | $(grep -o 'upload [0-9.]* ms, rest [0-9.]* ms' /tmp/e.txt | tail -1) | $(grep 'n=708600' /tmp/e.txt | grep -o 'pack.*') | $(grep 'n=10813' /tmp/e.txt | grep -o 'd2h.*')"
done; done
for sd in 1 0; do for i in 1 2 3; do HSA_ENABLE_SDMA=$sd sedef_amd/bin/sedef align generate -k 11 $d/genome.fa $d/one/bucket_0000 2>&1 >/dev/null | tr '\r' '\n' | grep -o "Finished BED.*in [0-9.]*s" | sed "s/.*in /sdma=$sd plain run: /"; done; done
