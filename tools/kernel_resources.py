#!/usr/bin/env python3
"""Compiles one .hip file with -Rpass-analysis=kernel-resource-usage and prints one row per kernel:
VGPRs, AGPRs, SGPRs, spills, scratch bytes, LDS bytes, occupancy (waves/SIMD).  No GPU needed.
usage: python tools/kernel_resources.py ofps_amd/csrc/almeida.hip [name-filter]"""
import re
import subprocess
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd.build import FLAGS, HIPCC  # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    p = subprocess.run([HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    rows, cur = [], None
    for ln in p.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\((?!anonymous).*", "", dem.replace("(anonymous namespace)::", "")).replace("void ", "").replace("ofps::", "")}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
    for r in rows:
        if flt and flt not in r["name"]:
            continue
        print(f"{r['name'][:58]:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
              f"{r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")


if __name__ == "__main__":
    main()
