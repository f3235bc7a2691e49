#!/usr/bin/env python3
"""hip_flow read-ahead decoder loop (cv-decoder's call, 1080p) for a rocprofv3 kernel / memory-copy trace: run from the tree under test
(python tools/fb_decoder_trace.py [frames]); works on the round-5 tree too."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.getcwd())
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0)
gc.collect(); gc.freeze(); gc.disable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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
run_fb(8)
t0 = time.perf_counter(); run_fb(n); print(f"{(time.perf_counter() - t0) / n * 1e3:.4f} ms per frame", flush=True)
