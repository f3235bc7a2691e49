#!/usr/bin/env python3
"""summary.json of the cfg3 chain profile: per (kernel, grid) average duration and calls per chain iteration from the
rocprofv3 kernel trace, plus the per-dispatch means of the PMC counters (VALU instructions per wave; FETCH_SIZE /
WRITE_SIZE in the units rocprofv3 reports on gfx950: KiB of 64-byte... see MI355X_MICROARCH.md -- kept raw here).
usage: cfg3_profile_summary.py <dir written by tools/cfg3_profile.sh> <iterations>"""
import collections, csv, glob, json, sys

root, iters = sys.argv[1], int(sys.argv[2])


def short(name):
    return name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").replace("ofps::", "")


out = {"what": "rocprofv3 of tools/prof_lk.py (cfg3 chain: LK flow -> densify 150x84 -> Almeida LSQ on 2.07 M records), %d iterations" % iters}
dur = collections.defaultdict(list)
for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(short(r["Kernel_Name"]), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = [{"kernel": k, "grid_x": g, "calls_per_iteration": round(len(v) / iters, 2), "avg_us": round(sum(v) / len(v) / 1e3, 2),
         "us_per_iteration": round(sum(v) / iters / 1e3, 1)} for (k, g), v in dur.items()]
rows.sort(key=lambda r: -r["us_per_iteration"])
out["kernel_time_sum_us_per_iteration"] = round(sum(r["us_per_iteration"] for r in rows), 1)
out["kernels"] = rows
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"]) + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pmc = {}
for k, c in sorted(acc.items()):
    row = {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}
    if row.get("SQ_WAVES"):
        row["valu_instr_per_wave"] = round(row.get("SQ_INSTS_VALU", 0) / row["SQ_WAVES"], 1)
    pmc[k] = row
out["pmc"] = pmc
print(json.dumps(out, indent=1))
