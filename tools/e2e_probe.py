#!/usr/bin/env python3
"""Where the PCIe-inclusive per-frame time goes (1080p, Decoder::process_frame shape)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
W, H, B, R = 1920, 1080, 16, 16
ctx = HipContext(0)
fr = synth.luma_sequence(4, W, H, max_step=16)
pins = [ctx.pinned_frame(H, W) for _ in range(3)]
ents = [ctx.pinned_array((8040, 4)) for _ in range(2)]
page = [np.empty((H, W), np.uint8) for _ in range(3)]
kw = dict(block=B, search_range=R, detector=False, estimator=False)
N = 300
def t(fn):
    fn(); t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) / N * 1e3
def copy_only(dst):
    def f():
        for k in range(N): np.copyto(dst[k % 3], fr[k % 4])
    return f
print("host copy into 3 pinned buffers   ms/frame", round(t(copy_only(pins)), 4))
print("host copy into 3 pageable buffers ms/frame", round(t(copy_only(page)), 4))
print("host copy into 1 pinned buffer    ms/frame", round(t(lambda: [np.copyto(pins[0], fr[k % 4]) for k in range(N)]), 4))
def sync_nocopy():
    ctx.reset_frames()
    for k in range(N): ctx.frame_wait(ctx.push_frame_async(pins[k % 3], out_entries=ents[0], **kw))
def async_run(copy):
    def f():
        ctx.reset_frames(); prev = None
        for k in range(N):
            if copy: np.copyto(pins[k % 3], fr[k % 4])
            tk = ctx.push_frame_async(pins[k % 3], out_entries=ents[k % 2], **kw)
            if prev is not None: ctx.frame_wait(prev)
            prev = tk
        ctx.frame_wait(prev)
    return f
def async_copy_after_push():
    # the decoder fills buffer k+1 AFTER frame k was pushed (what a read-ahead thread does): copy overlaps GPU work
    ctx.reset_frames(); prev = None
    np.copyto(pins[0], fr[0])
    for k in range(N):
        tk = ctx.push_frame_async(pins[k % 3], out_entries=ents[k % 2], **kw)
        np.copyto(pins[(k + 1) % 3], fr[(k + 1) % 4])
        if prev is not None: ctx.frame_wait(prev)
        prev = tk
    ctx.frame_wait(prev)
for k in range(3): np.copyto(pins[k], fr[k])
print("sync, frames already pinned        ms/frame", round(t(sync_nocopy), 4))
print("read-ahead, frames already pinned  ms/frame", round(t(async_run(False)), 4))
print("read-ahead, copy before push       ms/frame", round(t(async_run(True)), 4))
print("read-ahead, copy after push        ms/frame", round(t(async_copy_after_push), 4))
