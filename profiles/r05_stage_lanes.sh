#!/bin/bash
# Round 5: buckets of one process IN FLIGHT (SDF_BUCKET_LANES, host/pipeline.cc generate_many): eight chr1-sized buckets and the
# chr1-sized seed file cut into eight buckets, one process, 1 / 2 / 3 buckets at a time; outputs compared with the one-at-a-time run.
#   bash profiles/r05_stage_lanes.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r05lanes; mkdir -p $out
d=/tmp/sdf_stage_lanes
python3 - > $out/gen.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
from sedef_amd.host import build_host
build_host()
d = "/tmp/sdf_stage_lanes"; os.makedirs(d + "/align", exist_ok=True); os.makedirs(d + "/one", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 8, d + "/align", fa)
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa)
print(nseeds, sorted(os.listdir(d + "/align")))
PY
tail -1 $out/gen.log
B=sedef_amd/bin/sedef
now() { date +%s.%N; }
mkdir -p $d/big; for k in 0 1 2 3 4 5 6 7; do cp $d/one/bucket_0000 $d/big/bucket_000$k; done
for set in big align; do
  for lanes in 1 2 3 2 1; do
    rm -f $d/$set/*.aligned.bed
    t0=$(now); SDF_BUCKET_LANES=$lanes $B align generate -k 11 $d/genome.fa $d/$set > /dev/null 2> $out/${set}_$lanes.log; t1=$(now)
    echo "$set: 8 buckets, 1 process, $lanes in flight: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(grep -o 'All 8 buckets done in [0-9.]*s' $out/${set}_$lanes.log); stage clocks: $(grep -o 'Finished BED [^ ]* in [0-9.]*s' $out/${set}_$lanes.log | grep -o 'in [0-9.]*s' | tr '\n' ' ')"
    if [ $lanes = 1 ] && [ ! -d $d/$set.ref ]; then mkdir -p $d/$set.ref; cp $d/$set/*.aligned.bed $d/$set.ref/; fi
    ok=1; for b in $d/$set/bucket_????; do cmp -s $b.aligned.bed $d/$set.ref/$(basename $b).aligned.bed || ok=0; done; echo "   outputs equal to one at a time: $ok"
  done
done
