#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel from *_counter_collection.csv files under a directory.

usage: pmc_summary.py DIR [kernel-substring] [rows-per-launch]
Prints one line per (kernel, counter): total over launches, launches, and per-launch / per-row averages.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    tot = defaultdict(float)
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for rec in csv.DictReader(open(f)):
            name = rec["Kernel_Name"]
            if sub and sub not in name:
                continue
            short = name.split("(")[0]
            key = (short, rec["Counter_Name"])
            tot[key] += float(rec["Counter_Value"])
            disp[key].add(rec["Dispatch_Id"])
    for (k, c), v in sorted(tot.items()):
        n = len(disp[(k, c)])
        line = "%-40s %-18s total %.4e launches %d per_launch %.4e" % (k, c, v, n, v / n)
        if rows:
            line += " per_row %.2f" % (v / n / rows)
        print(line)


if __name__ == "__main__":
    main()
