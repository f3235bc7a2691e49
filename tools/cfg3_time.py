#!/usr/bin/env python3
"""cfg3 (BASELINE configs[2]) stage timings on one MI355X: 1080p pair -> 3-level LK flow -> per-pixel records ->
densify 150x84 -> Almeida LSQ on 2.07 M records.  HIP-event style wall timing over repeated calls; prints JSON."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4)


ctx = HipContext(0); ctx.use_torch_stream()
out = {}
for name, step in (("small_motion_pm3", 3), ("bench_motion_pm16", 16), ("camera_warp_subpixel", None)):
    fr = synth.luma_sequence(2, 1920, 1080, max_step=step, seed=11) if step else synth.camera_warp_pair()
    dfr = torch.from_numpy(fr).cuda()
    d_ent = torch.empty((1920 * 1080, 4), dtype=torch.float32, device="cuda")
    f84 = torch.empty((150 * 84, 2), dtype=torch.float32, device="cuda")
    q1 = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    lk = lambda: ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), 1920, 1080, 1920, 3, 4, 3, None, d_ent.data_ptr())
    den = lambda: ctx.densify_raster_dev(d_ent.data_ptr(), None, 1920, 1080, 150, 84, f84.data_ptr())     # per-pixel producer: rectangle walk
    den_generic = lambda: ctx.densify_dev(d_ent.data_ptr(), 1920 * 1080, 1, 150, 84, f84.data_ptr())     # any records: stable sort
    alm = lambda: ctx.almeida_dev(d_ent.data_ptr(), 1920 * 1080, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q1.data_ptr())
    def chain(): lk(); den(); alm()
    out[name] = {"lk_flow_ms": timeit(lk), "densify_150x84_ms": timeit(den), "densify_150x84_generic_sort_ms": timeit(den_generic), "almeida_lsq_2p07M_ms": timeit(alm), "chain_ms": timeit(chain),
                 "Mvectors_per_s_chain": None}
    out[name]["Mvectors_per_s_chain"] = round(1920 * 1080 / out[name]["chain_ms"] / 1e3, 1)
print(json.dumps(out, indent=1))
