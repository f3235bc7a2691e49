#!/usr/bin/env python3
"""Static view of the dense Almeida solve's SERIAL wave (VERDICT r5 "next" #9): compiles almeida.hip to gfx950 assembly, takes
almeida_lsq_cluster_kernel<true, 8, 1024> and counts, between the s_memtime phase stamps of OFPS_HIP_ALMEIDA_PROF (in program order:
slot 0 step start, 1 records done, 2 block sum done, 3 granule published + gathered, 5 serial wave starts the update, 6 rotation updated,
4 step end), the instructions and the SGPR-spill lane operations (v_writelane / v_readlane on the spill VGPR).  No GPU needed.
  python tools/almeida_spill_static.py"""
import bisect
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ofps_amd.build import FLAGS, HIPCC  # noqa: E402

with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "alm.s")
    subprocess.run([HIPCC] + [f for f in FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "ofps_amd", "csrc", "almeida.hip")],
                   check=True, capture_output=True)
    lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN4ofps26almeida_lsq_cluster_kernelILb1ELi8ELi1024")][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
stamps = [i for i, l in enumerate(body) if "s_memtime" in l]
lane = Counter()
for l in body:
    m = re.search(r"v_(writelane|readlane)_b32 (\S+), (\S+),", l)
    if m:
        lane[(m.group(2) if m.group(1) == "writelane" else m.group(3)).strip(",")] += 1
spill_vgpr = lane.most_common(1)[0][0]
names = ["prologue", "slot0->1 records", "slot1->2 block sum", "slot2->3 publish + gather", "slot3->5 wait for the serial wave's turn", "slot5->6 UPDATE (serial wave)",
         "slot6->4 barrier", "after the loop (solo finish, epilogue)"]
ins = [0] * (len(stamps) + 1); rd = [0] * (len(stamps) + 1); wr = [0] * (len(stamps) + 1)
for i, l in enumerate(body):
    k = bisect.bisect(stamps, i)
    if re.match(r"\s+(v_|s_|ds_|global_|buffer_|flat_)", l):
        ins[k] += 1
    if re.search(rf"v_readlane_b32 \S+, {spill_vgpr},", l):
        rd[k] += 1
    if re.search(rf"v_writelane_b32 {spill_vgpr},", l):
        wr[k] += 1
print(f"almeida_lsq_cluster_kernel<true, 8, 1024>: {len(body)} assembly lines, SGPR spill VGPR {spill_vgpr}: {sum(wr)} spills (v_writelane), {sum(rd)} reloads (v_readlane)")
print(f"{'segment (program order)':48s} {'instr':>6s} {'spills':>7s} {'reloads':>8s}")
for k in range(len(stamps) + 1):
    print(f"{names[k] if k < len(names) else '?':48s} {ins[k]:6d} {wr[k]:7d} {rd[k]:8d}")
