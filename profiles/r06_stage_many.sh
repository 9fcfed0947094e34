#!/bin/bash
# Round 6 (same script as round 5): the chr1-sized stage (BASELINE configs[2] shape) as 8 buckets -- one process per bucket (sedef.sh:187-190) against
# ONE process for all (sedef align generate genome.fa align/): wall time, stage clocks, outputs compared.   bash profiles/r05_stage_many.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06stage; mkdir -p $out
d=/tmp/sdf_stage_many
python3 - > $out/gen.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
from sedef_amd.host import build_host
build_host()
d = "/tmp/sdf_stage_many"; os.makedirs(d + "/align", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 8, d + "/align", fa)
print(nseeds, sorted(os.listdir(d + "/align")))
host.bucket(fa + ".seeds.bed", 1, d + "/one", fa) if os.makedirs(d + "/one", exist_ok=True) is None else None
PY
tail -1 $out/gen.log
B=sedef_amd/bin/sedef
now() { date +%s.%N; }
# one bucket holding everything, one process (round 4's figure)
for i in 1 2; do t0=$(now); $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/one.bed 2> $out/one_$i.log; t1=$(now); echo "one bucket, one process: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(grep -o 'Finished BED.*' $out/one_$i.log | cut -c1-90)"; done
# eight buckets, eight processes one after the other
for i in 1 2; do
  t0=$(now)
  for b in $d/align/bucket_????; do $B align generate -k 11 $d/genome.fa $b > $b.sep.bed 2> $out/sep_$(basename $b).log; done
  t1=$(now); echo "8 buckets, 8 processes: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; stage clocks: $(grep -ho 'in [0-9.]*s' $out/sep_bucket_*.log | tr '\n' ' ')"
done
# eight buckets, one process
mkdir -p $d/logs
for i in 1 2; do
  rm -f $d/align/*.aligned.bed
  t0=$(now); SDF_DEBUG_TIMING=${DBG:-} $B align generate -k 11 --log-dir $d/logs $d/genome.fa $d/align > /dev/null 2> $out/many_$i.log; t1=$(now)
  echo "8 buckets, 1 process: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(grep 'All 8 buckets' $out/many_$i.log); stage clocks: $(grep -ho 'in [0-9.]*s' $d/logs/*.log | tr '\n' ' ')"
done
ok=1; for b in $d/align/bucket_????; do cmp -s $b.sep.bed $b.aligned.bed || ok=0; done; echo "outputs equal: $ok; finished lines in logs: $(grep -l Finished $d/logs/*.log | wc -l)"
cat $d/align/*.aligned.bed | sort | sha256sum; sort $d/one.bed | sha256sum
# eight chr1-SIZED buckets (the whole seed file eight times under eight names): one process per bucket against one process
mkdir -p $d/big; for k in 0 1 2 3 4 5 6 7; do cp $d/one/bucket_0000 $d/big/bucket_000$k; done
t0=$(now); for b in $d/big/bucket_????; do $B align generate -k 11 $d/genome.fa $b > $b.sep.bed 2> $out/bigsep_$(basename $b).log; done; t1=$(now)
echo "8 chr1-sized buckets, 8 processes: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; stage clocks: $(grep -ho 'in [0-9.]*s' $out/bigsep_bucket_*.log | tr '\n' ' ')"
for i in 1 2; do
  rm -f $d/big/*.aligned.bed
  t0=$(now); $B align generate -k 11 $d/genome.fa $d/big > /dev/null 2> $out/bigmany_$i.log; t1=$(now)
  echo "8 chr1-sized buckets, 1 process: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(grep 'All 8 buckets' $out/bigmany_$i.log); stage clocks: $(grep -o 'Finished BED [^ ]* in [0-9.]*s' $out/bigmany_$i.log | grep -o 'in [0-9.]*s' | tr '\n' ' ')"
done
ok=1; for b in $d/big/bucket_????; do cmp -s $b.sep.bed $b.aligned.bed || ok=0; cmp -s $b.aligned.bed $d/one.bed || ok=0; done; echo "chr1-sized outputs equal (each process / one process / the single run): $ok"
