#!/usr/bin/env python3
"""One iteration's kernels in dispatch order from a rocprofv3 kernel_trace.csv: name, grid, duration, gap to the previous kernel.
  python tools/ktrace_seq.py <k_kernel_trace.csv> <first kernel substring> [which occurrence, default last]"""
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
i0 = starts[which]
i1 = starts[which + 1] if which + 1 < len(starts) and which != -1 else len(rows)
prev_end = None
tot = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{n[:40]:<40} grid {r.get('Grid_Size_X', '?'):>8} x {r.get('Grid_Size_Y', '?'):>5}  {(e - s) / 1e3:8.1f} us   gap {gap:6.1f}")
    tot += (e - s) / 1e3 + gap
    prev_end = e
print(f"span {tot:.1f} us")
