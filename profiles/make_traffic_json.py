#!/usr/bin/env python3
"""Writes profiles/hbm_traffic.json from the PMC summaries of one profiles/collect.sh run.

usage: make_traffic_json.py gpurun_out/<tag> <profiles/rNN_pair_kernel_pmc.txt>
Reads pmc_fetch.txt / pmc_write.txt / pmc_sq.txt (profiles/pmc_summary.py lines of the isolated extz2_pair_kernel launch:
separate --pmc passes, KB units, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md) and records next to the numbers
the sha256 of sedef_amd/csrc/extz2_pair.hip as it was when they were taken: bench.py prints the counters only while that
file is unchanged."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, kernel, counter):
    for line in open(path):
        if kernel in line and counter in line:
            return float(re.search(r"per_launch ([0-9.e+]+)", line).group(1))
    raise SystemExit("%s: no %s line for %s" % (path, counter, kernel))


def main():
    d, label = sys.argv[1], sys.argv[2]
    k = "extz2_pair_kernel<3"
    fetch_kb = per_launch(os.path.join(d, "pmc_fetch.txt"), k, "FETCH_SIZE")
    write_kb = per_launch(os.path.join(d, "pmc_write.txt"), k, "WRITE_SIZE")
    valu = per_launch(os.path.join(d, "pmc_sq.txt"), k, "SQ_INSTS_VALU")
    src = os.path.join(ROOT, "sedef_amd", "csrc", "extz2_pair.hip")
    out = {
        "bytes_per_step": (2.0 * fetch_kb + write_kb) * 1024.0,
        "fetch_kb_per_launch": fetch_kb, "write_kb_per_launch": write_kb,
        "workload": "configs[1], 100000 tasks, one extz2_pair_kernel<3> launch (SDF_PIPELINE=0)",
        "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KB units; FETCH doubled per the gfx950 "
                  "note of MI355X_MICROARCH.md)" % label,
        "valu_insts_per_step": valu,
        "valu_source": "%s (rocprofv3 --pmc SQ_INSTS_VALU, wavefront instructions of the same launch)" % label,
        "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
    }
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
