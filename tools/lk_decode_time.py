#!/usr/bin/env python3
"""Wall time of ofps_hip_lk_decode (hip_lk's Decoder::process_frame shape: two host frames in, down-sampled records out)
at 1080p, with and without cv-decoder's contrast mask."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0)
import gc
gc.collect(); gc.freeze(); gc.disable()        # a generation-2 pass of CPython's collector is 35-50 ms after `import torch`/numpy: it would land inside one of the 20-call loops
fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
for mask in (False, True):
    for _ in range(3): ctx.lk_decode(fr[0], fr[1], contrast_mask=mask)
    t0 = time.perf_counter()
    for _ in range(20): ent, grid = ctx.lk_decode(fr[0], fr[1], contrast_mask=mask)
    print(f"lk_decode 1080p -> {grid[0]}x{grid[1]}, contrast_mask={mask}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call, {len(ent)} records")
fr4 = synth.luma_sequence(4, 1920, 1080, max_step=3, seed=11)
for mask in (False, True):
    ctx.lk_reset(); ctx.lk_push_frame(fr4[0], contrast_mask=mask)
    for k in range(1, 4): ctx.lk_push_frame(fr4[k], contrast_mask=mask)
    t0 = time.perf_counter()
    for k in range(20): ctx.lk_push_frame(fr4[k % 4], contrast_mask=mask)
    print(f"lk_push_frame 1080p (stream form, one upload), contrast_mask={mask}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per frame")
# read-ahead form: frames in page-locked memory, two tickets in flight (the upload of frame k+1 beside the flow of pair k-1, k)
pins = [ctx.pinned_frame(1080, 1920) for _ in range(4)]
for k in range(4): np.copyto(pins[k], fr4[k])
out = [np.zeros((150 * 150, 4), np.float32) for _ in range(2)]     # capacity: min(max_w, W) * min(max_h, H) records
for mask in (False, True):
    ctx.lk_reset()
    def run(n):
        prev = None
        for k in range(n):
            t = ctx.lk_push_frame_async(pins[k % 4], contrast_mask=mask)
            if prev is not None: ctx.lk_frame_wait(prev, out[k & 1])
            prev = t
        ctx.lk_frame_wait(prev, out[n & 1])
    run(8)
    t0 = time.perf_counter(); run(200)
    print(f"lk_push_frame_async + lk_frame_wait 1080p (read-ahead, 2 tickets), contrast_mask={mask}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per frame")
# hip_flow (OFPS_HIP_FLOW_FARNEBACK, cv-decoder's arguments: levels 5, winsize 13, 3 iterations): pair call / stream form / read-ahead form
FB = dict(levels=5, radius=6, iters=3, contrast_mask=True, farneback=True, use_previous=True)     # cv-decoder's call incl. OPTFLOW_USE_INITIAL_FLOW from the second pair on
for _ in range(3): ctx.lk_decode(fr[0], fr[1], **FB)
t0 = time.perf_counter()
for _ in range(20): ent, grid = ctx.lk_decode(fr[0], fr[1], **FB)
print(f"hip_flow lk_decode 1080p -> {grid[0]}x{grid[1]} (pair call: both frames through the pyramid + expansion): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call, {len(ent)} records")
ctx.lk_reset(); ctx.lk_push_frame(fr4[0], **FB)
for k in range(1, 4): ctx.lk_push_frame(fr4[k], **FB)
h0 = ctx.flow_cache_hits()
t0 = time.perf_counter()
for k in range(20): ctx.lk_push_frame(fr4[k % 4], **FB)
print(f"hip_flow lk_push_frame 1080p (stream form): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per frame, {ctx.flow_cache_hits() - h0} of 20 frames reused the previous frame's expansion")
ctx.lk_reset()
def run_fb(n):
    prev = None
    for k in range(n):
        t = ctx.lk_push_frame_async(pins[k % 4], **FB)
        if prev is not None: ctx.lk_frame_wait(prev, out[k & 1])
        prev = t
    ctx.lk_frame_wait(prev, out[n & 1])
run_fb(8)
h0 = ctx.flow_cache_hits()
t0 = time.perf_counter(); run_fb(200)
print(f"hip_flow lk_push_frame_async + lk_frame_wait 1080p (read-ahead, 2 tickets): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per frame, {ctx.flow_cache_hits() - h0} of 200 reused")
