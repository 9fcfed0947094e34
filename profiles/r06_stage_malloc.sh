#!/bin/bash
# Round 6: does the stage pay for page faults of fresh memory?  The chr1-sized bucket with glibc's malloc on huge pages /
# with large blocks kept in the heap.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06malloc; mkdir -p $out
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; ldd --version | head -1
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
run() {  # label, env...
  label=$1; shift
  for i in 1 2 3; do
    t0=$(now); env "$@" $B align generate -k 11 $d/genome.fa $d/one/bucket_0000 > $d/out_$label.bed 2> $out/$label.$i.log; t1=$(now)
    echo "$label: wall $(awk "BEGIN{printf \"%.2f\", $t1 - $t0}") s; $(tr '\r' '\n' < $out/$label.$i.log | grep -o 'Finished BED.*' | grep -o 'in [0-9.]*s'); $(grep -o 'process:.*' $out/$label.$i.log); sha $(sha256sum < $d/out_$label.bed | cut -c1-12)"
  done
}
run base SDF_DEBUG_TIMING=1
run thp SDF_DEBUG_TIMING=1 GLIBC_TUNABLES=glibc.malloc.hugetlb=1
run heap SDF_DEBUG_TIMING=1 MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456
run both SDF_DEBUG_TIMING=1 GLIBC_TUNABLES=glibc.malloc.hugetlb=1 MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456
grep -h "sdf_reserve\|sdf_create device" $out/base.1.log $out/thp.1.log
