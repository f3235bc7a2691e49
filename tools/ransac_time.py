import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((120, 67), (150, 84)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    f = lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, True, 200, 0.05, 1000, 7, q.data_ptr())
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print(f"ransac n={n}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms  q={q.cpu().numpy().ravel()}")
