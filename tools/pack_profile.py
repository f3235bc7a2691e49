#!/usr/bin/env python3
"""pack_profile.py <gpurun_out profile dir> <profiles/rNN/<name>>: the compact, committed form of one tools/prof_counters.sh /
tools/cfg3_profile.sh run -- rocprofv3's kernel_stats.csv as is, every PMC pass aggregated per (kernel, counter) to
dispatches / mean / min / max, plus the scripts' own summary files."""
import collections
import csv
import glob
import os
import shutil
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[(r["Kernel_Name"][:80], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        with open(os.path.join(dst, os.path.basename(d) + ".csv"), "w", newline="") as out:
            w = csv.writer(out)
            w.writerow(["kernel", "grid", "counter", "dispatches", "mean_per_dispatch", "min", "max"])
            for (k, c, g), v in sorted(acc.items()):
                w.writerow([k, g, c, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
    for name in ("summary.txt", "summary.json", "stage_times.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, name))


if __name__ == "__main__":
    main()
