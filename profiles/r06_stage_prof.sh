# Kernel trace of the product CLI on the chr1-sized bucket (the stage's device side, kernel by kernel).   bash profiles/r06_stage_prof.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
out=gpurun_out/r06stageprof; mkdir -p $out
python3 - > $out/gen.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hostgen
from sedef_amd import host
from sedef_amd.host import build_host
build_host()
d = "/tmp/sdf_stage_prof"; os.makedirs(d + "/buckets", exist_ok=True)
fa = d + "/genome.fa"
genome, nseeds = hostgen.make_chr1_genome(fa)
host.bucket(fa + ".seeds.bed", 1, d + "/buckets", fa)
print(nseeds, os.listdir(d + "/buckets"))
PY
bed=$(ls /tmp/sdf_stage_prof/buckets/* | head -1)
for i in 1 2; do sedef_amd/bin/sedef align generate -k 11 /tmp/sdf_stage_prof/genome.fa $bed > /tmp/sdf_stage_prof/out.bed 2> $out/plain_$i.log; grep Finished $out/plain_$i.log; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o stage -- sedef_amd/bin/sedef align generate -k 11 /tmp/sdf_stage_prof/genome.fa $bed > /tmp/sdf_stage_prof/out2.bed 2> $out/prof.log
grep Finished $out/prof.log
cmp /tmp/sdf_stage_prof/out.bed /tmp/sdf_stage_prof/out2.bed && echo same output
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/stage_kernel_stats.csv
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/stage_kernel_trace.csv
rm -rf $out/prof
head -30 $out/stage_kernel_stats.csv | cut -c1-200
python3 - <<'PY'
import csv
rows = []
for r in csv.DictReader(open("gpurun_out/r06stageprof/stage_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], r.get("Queue_Id", "")))
rows.sort()
# skip the warm-up anchors call of sdf_reserve: start at the first ref_keys kernel after a gap
t0 = [s for s, e, k, q in rows if "ref_keys" in k][-1]
with open("gpurun_out/r06stageprof/timeline.txt", "w") as f:
    for s, e, k, q in rows:
        if s >= t0 and (e - s) > 20000:
            f.write("%9.3f %9.3f  %7.3f ms  q%s  %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k))
print(open("gpurun_out/r06stageprof/timeline.txt").read())
PY
