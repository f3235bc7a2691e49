#!/usr/bin/env python3
"""Repeated timing of the default Almeida LSQ path on the cfg3 sizes (ms per estimate; 5 repeats of 50 calls)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((120, 67), (150, 84), (480, 270), (960, 540), (1920, 1080)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    f = lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())
    res = []
    for r in range(5):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e3)
    print(f"n={n}: " + " ".join(f"{x:.4f}" for x in res), flush=True)
