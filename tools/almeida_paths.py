#!/usr/bin/env python3
"""A/B of the two LSQ drivers (one-workgroup kernel vs one launch per step) over batch sizes at N = 8,040."""
import json, os, sys, time
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
out = {}
for n_side in ((120, 67), (64, 36), (32, 18)):
    n = n_side[0] * n_side[1]
    for batch in (1, 2, 4, 8, 16, 64, 256):
        e = np.stack([synth.rotation_field(*n_side, seed=k) for k in range(min(batch, 4))])
        e = np.concatenate([e] * ((batch + 3) // 4))[:batch]
        d = torch.from_numpy(np.ascontiguousarray(e)).cuda()
        q = torch.empty((batch, 4), dtype=torch.float32, device="cuda")
        out[f"n{n}_b{batch}"] = round(timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, batch, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())), 4)
print(os.environ.get("OFPS_HIP_ALMEIDA_PATH", "default"), json.dumps(out))
