#!/usr/bin/env python3
"""Phase profile of the cluster Almeida solver (OFPS_HIP_ALMEIDA_PROF=1 prints the in-kernel s_memtime table) and
timing of EPT variants for the dense sizes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((120, 67), (150, 84), (480, 270), (960, 540), (1920, 1080)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    for ept in (1, 2, 4, 8):
        if (n + ept * 1024 - 1) // (ept * 1024) > 256: continue
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster"); ctx.set_option("OFPS_HIP_ALMEIDA_EPT", ept)
        ctx.set_option("OFPS_HIP_ALMEIDA_PROF", None)
        ms = timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr()))
        print(f"n={n} ept={ept}: {ms:.4f} ms", file=sys.stderr, flush=True)
        ctx.set_option("OFPS_HIP_ALMEIDA_PROF", 1)
        ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())
        torch.cuda.synchronize()
