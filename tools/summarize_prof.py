#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs written by tools/prof_counters.sh into a short text summary
(per-kernel durations from the kernel trace, per-dispatch mean of every PMC counter)."""
import csv, glob, os, sys, collections

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
kfilter = sys.argv[2] if len(sys.argv) > 2 else ""

for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, root))
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 12:
            print("  ", {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("== kernel trace:", os.path.relpath(f, root))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
        print(f"   {k[:90]:90s} calls={len(v):4d} avg_us={sum(v)/len(v)/1e3:10.2f} min_us={min(v)/1e3:10.2f}")
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if kfilter and kfilter not in row["Kernel_Name"]:
            continue
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters:", os.path.relpath(f, root))
    for k, cs in agg.items():
        if not any("sad" in k or "almeida" in k or "densify" in k or "detect" in k or "sort" in k or "cell" in k for _ in [0]):
            continue
        for c, v in cs.items():
            print(f"   {k[:60]:60s} {c:24s} mean/dispatch={sum(v)/len(v):16.1f} n={len(v)}")
