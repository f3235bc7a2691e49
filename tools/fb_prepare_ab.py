#!/usr/bin/env python3
"""hip_flow read-ahead decoder (cv-decoder's call, 1080p): the new frame's pyramid + expansion on the upload's stream when the frame is pushed
(OFPS_HIP_FB_PREPARE_AHEAD=1, the default) against inside the pair's flow on the compute stream (0, round 5's order).  ms per frame, 5 x 100 frames."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0)
gc.collect(); gc.freeze(); gc.disable()
fr4 = synth.luma_sequence(4, 1920, 1080, max_step=3, seed=11)
pins = [ctx.pinned_frame(1080, 1920) for _ in range(4)]
for k in range(4): np.copyto(pins[k], fr4[k])
out = [np.zeros((150 * 150, 4), np.float32) for _ in range(2)]
FB = dict(levels=5, radius=6, iters=3, contrast_mask=True, farneback=True, use_previous=True)
def run_fb(n):
    prev = None
    for k in range(n):
        t = ctx.lk_push_frame_async(pins[k % 4], **FB)
        if prev is not None: ctx.lk_frame_wait(prev, out[k & 1])
        prev = t
    ctx.lk_frame_wait(prev, out[n & 1])
for rnd in range(3):
    for mode in (1, 0):
        ctx.set_option("OFPS_HIP_FB_PREPARE_AHEAD", mode)
        ctx.lk_reset(); run_fb(8)
        res = []
        for _ in range(5):
            t0 = time.perf_counter(); run_fb(100); res.append((time.perf_counter() - t0) / 100 * 1e3)
        print(f"prepare_ahead={mode}: " + " ".join(f"{x:.4f}" for x in res) + f"  median {sorted(res)[2]:.4f}", flush=True)
