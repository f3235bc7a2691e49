#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter values per dispatch and the derived VALU-busy
fraction (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)).  usage: pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").replace("ofps::", "")
            acc[name + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    row = {n: round(v) for n, v in m.items()}
    row["dispatches"] = len(next(iter(c.values())))
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        row["valu_busy_frac"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"] > 0:
        row["valu_insts_per_wave"] = round(m["SQ_INSTS_VALU"] / m["SQ_WAVES"], 1)
    out[k] = row
print(json.dumps(out, indent=1))
