#!/usr/bin/env python3
"""Soak of the hip_flow read-ahead decoder (Farneback with cv-decoder's arguments, OFPS_HIP_FLOW_USE_PREVIOUS): N frames of a 640 x 360
stream through ofps_hip_lk_push_frame_async / _frame_wait with two tickets in flight -- every frame reusing the previous frame's
pyramid + expansion and starting from the previous pair's flow -- compared bit for bit with the same stream run synchronously on a
second context that does nothing else; on the context under test unrelated work in between (densify calls, SAD searches, LK pair
calls, Farneback pair calls -- the last two use the same workspaces) and a restart every few hundred frames.  usage: flow_soak.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
W, H = 640, 360
ctx, ref = HipContext(0), HipContext(0)
fr = synth.luma_sequence(16, W, H, max_step=3, seed=77)
pins = [ctx.pinned_frame(H, W) for _ in range(3)]
KW = dict(levels=5, radius=6, iters=3, contrast_mask=True, farneback=True, use_previous=True)
rng = np.random.default_rng(2)
# negative control: the replay starts every pair from zero flow -- the comparison must see that (every pair after a stream's first differs)
KW_REF = dict(KW, use_previous=False) if os.environ.get("SOAK_NEGATIVE_CONTROL") else KW
bad = 0
t0 = time.perf_counter()
pending = []                                        # (ticket, expected)
def collect():
    global bad
    t, e = pending.pop(0)
    r = ctx.lk_frame_wait(t)
    if (r is None) != (e is None) or (r is not None and not np.array_equal(r[0].view(np.uint32), e[0].view(np.uint32))): bad += 1
restarts = 0
for k in range(N):
    if k and k % 331 == 0:                          # a restart: tickets collected, both streams forget frame, expansion and flow
        while pending: collect()
        ctx.lk_reset(); ref.lk_reset(); restarts += 1
    np.copyto(pins[k % 3], fr[k % 16])
    t = ctx.lk_push_frame_async(pins[k % 3], **KW)
    pending.append((t, ref.lk_push_frame(fr[k % 16], **KW_REF)))
    if len(pending) > 1: collect()
    m = k % 11
    if m == 3:
        n = 50 + k % 400
        e = np.zeros((n, 4), np.float32); e[:, 0] = (np.arange(n) % 16 + 0.5) / 16; e[:, 1] = (np.arange(n) // 16 % 9 + 0.5) / 9
        e[:, 2:] = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        ctx.densify(e, 16, 9)
    elif m == 5: ctx.sad_flow(fr[k % 16], fr[(k + 1) % 16], 16, 8)
    elif m == 7: ctx.lk_decode(fr[(k + 3) % 16], fr[(k + 4) % 16], 3, 4, 3, contrast_mask=True)
    elif m == 9: ctx.farneback_flow(fr[(k + 5) % 16][:180, :320].copy(), fr[(k + 6) % 16][:180, :320].copy(), levels=int(rng.integers(1, 5)))
while pending: collect()
print(f"flow soak: {N} frames {W}x{H} in {time.perf_counter() - t0:.1f} s, {restarts} restarts, mismatching frames {bad}, "
      f"frames that reused the previous expansion {ctx.flow_cache_hits()} (reference context {ref.flow_cache_hits()})")
sys.exit(1 if bad else 0)
