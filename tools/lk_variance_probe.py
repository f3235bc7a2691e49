#!/usr/bin/env python3
"""LK flow on the +-3 px content (the perf gate's noisiest row): one process, event-timed medians like the cfg3 leg; run several fresh
processes of it with and without OFPS_HIP_LK_SERIAL=1 to see whether the one-launch pyramid's tile waits carry the spread."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
import bench_legs
W, H = 1920, 1080
ctx = HipContext(0); ctx.use_torch_stream()
d_ent = torch.empty((W * H, 4), dtype=torch.float32, device="cuda")
out = []
for step in (3, 16):
    fr = synth.luma_sequence(2, W, H, max_step=step, seed=11)
    dfr = torch.from_numpy(fr).cuda()
    f = lambda: ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), W, H, W, 3, 4, 3, None, d_ent.data_ptr())
    out.append(f"pm{step} {bench_legs._event_ms(ctx, f, 60):.4f}")
print(os.environ.get("OFPS_HIP_LK_SERIAL", "0"), " ".join(out), flush=True)
