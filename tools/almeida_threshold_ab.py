import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((32, 20), (32, 28), (32, 32), (40, 32), (50, 32), (64, 32), (64, 40)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    f = lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())
    row = [f"n={n}"]
    for path in ("wg", "cluster"):
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", path)
        res = []
        for r in range(3):
            for _ in range(5): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): f()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50 * 1e3)
        row.append(f"{path} {min(res):.4f}")
    ctx.set_option("OFPS_HIP_ALMEIDA_PATH", None)
    print("  ".join(row), flush=True)
