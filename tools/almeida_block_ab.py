import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ctx = HipContext(0); ctx.use_torch_stream()
for (w, h) in ((120, 67), (150, 84), (240, 135), (480, 270), (1920, 1080)):
    n = w * h
    d = torch.from_numpy(synth.rotation_field(w, h)).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    for block, ept in ((1024, 0), (256, 1), (256, 2)):
        if block == 256 and (n + ept * 256 - 1) // (ept * 256) > 256: continue
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster")
        ctx.set_option("OFPS_HIP_ALMEIDA_BLOCK", block)
        ctx.set_option("OFPS_HIP_ALMEIDA_EPT", ept or None)
        ctx.set_option("OFPS_HIP_ALMEIDA_PROF", None)
        f = lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())
        ms = timeit(f)
        print(f"n={n} block={block} ept={ept or 'auto'}: {ms:.4f} ms  q={q.cpu().numpy().ravel()}", file=sys.stderr, flush=True)
        ctx.set_option("OFPS_HIP_ALMEIDA_PROF", 1)
        f(); torch.cuda.synchronize()
