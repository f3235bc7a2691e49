#!/usr/bin/env python3
"""hip_flow read-ahead decoder: is a new frame's pyramid + expansion (upload stream) running BESIDE the previous pair's flow (compute stream)?
Reads rocprofv3 kernel_trace.csv (+ memory_copy_trace.csv when present) of tools/lk_decode_time.py, and prints one period of the last
loop: every kernel / copy with start and end relative to the pair's first flow kernel and the queue it ran on, plus the per-frame period.
  python tools/fb_overlap_trace.py <dir with k_kernel_trace.csv>"""
import csv
import glob
import os
import sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [dict(r, what="k") for r in csv.DictReader(open(kt))]
mc = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append({"Kernel_Name": "copy " + r.get("Direction", "?"), "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"], "Queue_Id": "dma", "what": "c"})
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "fb_area_kernel" in r["Kernel_Name"]]
per = [(int(rows[starts[k + 1]]["Start_Timestamp"]) - int(rows[starts[k]]["Start_Timestamp"])) / 1e3 for k in range(len(starts) - 61, len(starts) - 21)]
print(f"period between consecutive pairs' first flow kernel, 40 frames of the last loop: median {sorted(per)[len(per) // 2]:.1f} us, min {min(per):.1f}, max {max(per):.1f}")
i0, i1 = starts[-30], starts[-29]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 6:i1 + 2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ofps::", "").split("(")[0]
    print(f"{n[:34]:<34} queue {r.get('Queue_Id', '?'):>3}  start {(s - t0) / 1e3:8.1f}  end {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f} us)")
