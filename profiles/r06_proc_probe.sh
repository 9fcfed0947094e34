#!/bin/bash
# Round 6: where one `sedef align generate` PROCESS spends its wall time around the stage (chr1-sized bucket: stage clock 0.08 s,
# process 0.5-0.75 s), and what the size of the direction-flag workspace the stage reserves (16 GiB) has to do with it.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06proc; mkdir -p $out
d=/tmp/sdf_stage_one
python3 - > $out/gen.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
d = "/tmp/sdf_stage_one"; os.makedirs(d + "/one", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa)
PY
B=sedef_amd/bin/sedef
now() { date +%s.%N; }
run() {
  label=$1; shift
  for i in 1 2 3 4; do
    t0=$(now); env SDF_DEBUG_TIMING=1 "$@" $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_$label.bed 2> $out/$label.$i.log; t1=$(now)
    echo "$label: wall $(awk "BEGIN{printf \"%.3f\", $t1 - $t0}") s; $(tr '\r' '\n' < $out/$label.$i.log | grep -o 'Finished BED.*' | grep -o 'in [0-9.]*s'); $(grep -o 'sdf_create device 0: [0-9.]* ms' $out/$label.$i.log | head -1); $(grep -o 'sdf_reserve tasks.*' $out/$label.$i.log | grep -o '[0-9.]* ms' | head -1) reserve; $(grep -o 'process: .*' $out/$label.$i.log); sha $(sha256sum < $d/out_$label.bed | cut -c1-12)"
  done
}
for k in 1 2; do
run ws16 SDF_X=0
run ws8 SDF_STAGE_WS_GIB=8
run ws6 SDF_STAGE_WS_GIB=6
run ws4 SDF_STAGE_WS_GIB=4
done
# the loader's share
LD_DEBUG=statistics $B help 2>&1 | grep -i "total startup\|relocation\|load" | head -5
t0=$(now); $B help > /dev/null 2>&1; t1=$(now); echo "sedef help: wall $(awk "BEGIN{printf \"%.3f\", $t1 - $t0}") s"
