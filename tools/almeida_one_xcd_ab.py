#!/usr/bin/env python3
"""Cluster Almeida solver on block-vector sized fields: the flat write-through exchange (OFPS_HIP_ALMEIDA_ONE_XCD=0) against the
one-XCD launch (1, the default): ms per estimate (best of 5 x 50 calls), the in-kernel phase table, and whether the two
quaternions carry the same bits."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((24, 24), (40, 22), (64, 32), (80, 45), (120, 67), (150, 84), (240, 135)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    f = lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())
    row, bits = [f"n={n:6d}"], []
    for mode in (0, 1):
        ctx.set_option("OFPS_HIP_ALMEIDA_ONE_XCD", mode)
        res = []
        for r in range(5):
            for _ in range(5): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): f()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e3)
        bits.append(q.cpu().numpy().view(np.uint32).copy())
        row.append(f"one_xcd={mode}: {min(res):.4f} ms")
        if n in (8040, 12600):
            ctx.set_option("OFPS_HIP_ALMEIDA_PROF", 1); f(); torch.cuda.synchronize(); ctx.set_option("OFPS_HIP_ALMEIDA_PROF", None)
    row.append("same bits" if (bits[0] == bits[1]).all() else f"BITS DIFFER {bits}")
    ctx.set_option("OFPS_HIP_ALMEIDA_ONE_XCD", None)
    print("  ".join(row), flush=True)
