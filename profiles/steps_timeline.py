import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = []
for r in rows:
    name = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "reset_results" in name:
        steps.append({"t0": s, "k": []})
    if steps:
        steps[-1]["k"].append((name, s, e))
for i, st in enumerate(steps):
    t0 = st["t0"]
    out = []
    for key in ("strip_chain", "stripe_kernel", "extz2_strip_kernel", "extz2_lane_kernel", "traceback_kernel<3", "traceback_kernel<6", "traceback_kernel<5", "cigar_compact"):
        ks = [(s, e) for n, s, e in st["k"] if key in n]
        if ks:
            out.append("%s %.2f-%.2f" % (key.replace("extz2_", "").replace("_kernel", ""), (min(s for s, e in ks) - t0) / 1e6, (max(e for s, e in ks) - t0) / 1e6))
    print("step %d: " % i + " | ".join(out))
