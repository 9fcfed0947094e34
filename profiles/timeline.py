#!/usr/bin/env python3
"""Prints the kernel timeline (start/end in ms relative to the first kernel of the last batch) from a
rocprofv3 --kernel-trace --output-format csv directory.  usage: timeline.py DIR [last-N-kernels]"""
import csv, glob, os, sys
d = sys.argv[1]
lastn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")))
rows.sort()
rows = rows[-lastn:]
t0 = rows[0][0]
for s, e, k, q in rows:
    print("%9.3f %9.3f  %7.3f ms  q%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k))
