#!/usr/bin/env python3
"""Soak of the hip_lk read-ahead decoder: N frames of a 640 x 360 stream through ofps_hip_lk_push_frame_async / _frame_wait (two
tickets in flight, the one-launch pyramid's epoch advancing every frame), every result compared bit for bit with the pair call on
a second context, with unrelated work (generic densify calls, whose tables share scratch slots with other stages) in between.
usage: lk_soak.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
W, H = 640, 360
ctx, ref = HipContext(0), HipContext(0)
fr = synth.luma_sequence(16, W, H, max_step=3, seed=99)
pins = [ctx.pinned_frame(H, W) for _ in range(3)]
rng = np.random.default_rng(1)
want = {}
def pair(a, b):
    if (a, b) not in want: want[(a, b)] = ref.lk_decode(fr[a], fr[b])[0]
    return want[(a, b)]
bad = 0
t0 = time.perf_counter()
prev_t = None
for k in range(N):
    np.copyto(pins[k % 3], fr[k % 16])
    t = ctx.lk_push_frame_async(pins[k % 3])
    if prev_t is not None:
        r = ctx.lk_frame_wait(prev_t[0])
        j = prev_t[1]
        if j == 0: assert r is None
        elif not np.array_equal(r[0].view(np.uint32), pair((j - 1) % 16, j % 16).view(np.uint32)): bad += 1
    prev_t = (t, k)
    if k % 7 == 3:                                          # other stages' scratch traffic on the same context
        n = 50 + k % 400
        e = np.zeros((n, 4), np.float32); e[:, 0] = (np.arange(n) % 16 + 0.5) / 16; e[:, 1] = (np.arange(n) // 16 % 9 + 0.5) / 9
        e[:, 2:] = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        ctx.densify(e, 16, 9)
r = ctx.lk_frame_wait(prev_t[0])
if not np.array_equal(r[0].view(np.uint32), pair((N - 2) % 16, (N - 1) % 16).view(np.uint32)): bad += 1
print(f"lk soak: {N} frames {W}x{H} in {time.perf_counter() - t0:.1f} s, mismatching frames {bad}, tiles computed by a waiting child {ctx.lk_helped_tiles()}")
sys.exit(1 if bad else 0)
