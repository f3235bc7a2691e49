#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv compactly: name, calls, average / min / max us, share."""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{n[:44]:<44} {r['Calls']:>5}  avg {float(r['AverageNs']) / 1e3:8.1f}  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f} us  {r['Percentage']:>6} %")
