#!/usr/bin/env python3
"""Cluster Almeida solver: workgroup size x records per thread for block-vector sized fields (ms per estimate).
OFPS_HIP_ALMEIDA_BLOCK / _EPT select the variant; `auto` is what lsq_cluster picks by itself."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext


def timeit(fn, n=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((24, 24), (32, 32), (64, 32), (80, 45), (120, 67), (150, 84), (240, 135), (320, 180), (480, 270), (960, 540), (1920, 1080)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    row = [f"n={n:8d}"]
    ctx.set_option("OFPS_HIP_ALMEIDA_BLOCK", None); ctx.set_option("OFPS_HIP_ALMEIDA_EPT", None)
    row.append(f"auto {timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())):.4f}")
    if n <= 8192:
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "wg")
        row.append(f"wg {timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())):.4f}")
    if (16 <= (n + 255) // 256 <= 128 and n <= 65536) or 100000 < n < 600000:
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster"); ctx.set_option("OFPS_HIP_ALMEIDA_HIER", 0)
        row.append(f"auto/flat {timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())):.4f}")
        ctx.set_option("OFPS_HIP_ALMEIDA_HIER", 2)
        row.append(f"auto/2lvl {timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())):.4f}")
        ctx.set_option("OFPS_HIP_ALMEIDA_HIER", None)
    for block in (1024, 256):
        for ept in (1, 2, 4, 8):
            per = block * ept
            nb = (n + per - 1) // per
            if nb > 256 or (block == 256 and (ept > 4 or nb > 64)) or (n <= 65536 and ept > 4) or (block == 1024 and nb < 2 and ept > 1): continue
            ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster"); ctx.set_option("OFPS_HIP_ALMEIDA_BLOCK", block); ctx.set_option("OFPS_HIP_ALMEIDA_EPT", ept)
            ms = timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr()))
            row.append(f"{block}x{ept}({nb}) {ms:.4f}")
    ctx.set_option("OFPS_HIP_ALMEIDA_PATH", None)
    print("  ".join(row), flush=True)
