#!/usr/bin/env python3
"""Exhaustive vs pruned (partial-distortion elimination) SAD search on two kinds of content: the bench sequence
(independent +-16 jumps per 64x64 region: blocks straddle motion discontinuities) and a camera-motion sequence (one
global translation per frame + sensor noise, the decoder's real input).  Same output bits either way."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

ctx = HipContext(0); ctx.use_torch_stream()
W, H, B, R, P = 1920, 1080, 16, 16, 32
nb = (W // B) * (H // B)
out = {}
for name, kw in {"bench_sequence": dict(max_step=16), "camera_motion": dict(max_step=12, region=4096, noise=2),
                 "camera_motion_noise1": dict(max_step=12, region=4096, noise=1), "static_noise2": dict(max_step=0, noise=2)}.items():
    fr = synth.luma_sequence(P + 1, W, H, seed=7, **kw)
    d = torch.from_numpy(fr).cuda()
    o0 = torch.empty((P, nb, 3), dtype=torch.int32, device="cuda"); o1 = torch.empty_like(o0)
    e = torch.empty((P, nb, 4), dtype=torch.float32, device="cuda")
    res = {}
    for mode, ob in ((ctx.SAD_EXHAUSTIVE, o0), (ctx.SAD_PRUNED, o1)):
        ctx.set_sad_mode(mode)
        ms = timeit(lambda: ctx.sad_flow_dev(d.data_ptr(), P + 1, W, H, W, W * H, 0, B, R, e.data_ptr(), ob.data_ptr()))
        res["pruned" if mode else "exhaustive"] = {"ms_per_pair": round(ms / P, 5), "Mvectors_per_s": round(P * nb / ms / 1e3, 1)}
    res["overflow_strips_of"] = [ctx.sad_pruned_overflow_strips(), P * 15 * 67]
    res["identical"] = bool((o0 == o1).all().item())
    res["speedup"] = round(res["exhaustive"]["ms_per_pair"] / res["pruned"]["ms_per_pair"], 2)
    out[name] = res
ctx.set_sad_mode(ctx.SAD_EXHAUSTIVE)
print(json.dumps(out, indent=1))
