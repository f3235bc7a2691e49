#!/usr/bin/env python3
"""Soak of the fused per-frame path (ofps_hip_push_frame_async / ofps_hip_frame_wait: search + detector + estimator per frame, two
tickets in flight) with OTHER entry points called on the same context between the pushes -- pair searches of another geometry,
generic densify / detect / Almeida calls, dense flows, the hip_lk decoder: stages that share scratch slots with the stream's own
stages.  Every frame's vectors, island and quaternion are compared with an undisturbed second context.  usage: pipeline_soak.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
RANSAC = bool(int(os.environ.get("SOAK_RANSAC", "0")))       # the stream's estimator: the reference's default RANSAC instead of LSQ
NEG = int(os.environ.get("SOAK_NEGATIVE_CONTROL", "0"))     # 1: compare with the NEXT frame's expectation -- every frame must then mismatch
W, H, B, R = 640, 360, 16, 16
ctx, ref = HipContext(0), HipContext(0)
fr = synth.luma_sequence(12, W, H, max_step=8, seed=5)
small = synth.luma_sequence(2, 320, 176, max_step=4, seed=6)
rot = synth.rotation_field(64, 36)
rng = np.random.default_rng(2)
want = {}
def expect(j):
    """undisturbed: the synchronous call on the other context, frame j after frame j - 1"""
    k = j % 12
    if k not in want:
        ref.reset_frames()
        ref.push_frame(fr[(k - 1) % 12])
        want[k] = ref.push_frame(fr[k], block=B, search_range=R, want_entries=True, use_ransac=RANSAC, num_iters=50, num_samples=300, seed=7)
    return want[k]
nb = (W // B) * (H // B)
pins = [ctx.pinned_frame(H, W) for _ in range(3)]
outs = [ctx.pinned_array((nb, 4)) for _ in range(3)]
bad = 0
def check(j, res, ent):
    global bad
    if j == 0:
        bad += bool(res["have_vectors"]); return
    e = expect(j + NEG)
    ok = res["have_vectors"] and np.array_equal(ent.view(np.uint32), e["entries"].view(np.uint32)) and \
         (res["motion"] is None) == (e["motion"] is None) and (res["motion"] is None or res["motion"][0] == e["motion"][0]) and \
         np.array_equal(res["quat"].view(np.uint32), e["quat"].view(np.uint32))
    bad += not ok
t0 = time.perf_counter()
prev = None
for k in range(N):
    np.copyto(pins[k % 3], fr[k % 12])
    t = ctx.push_frame_async(pins[k % 3], block=B, search_range=R, out_entries=outs[k % 3], use_ransac=RANSAC, num_iters=50, num_samples=300, seed=7)
    which = k % 6                                            # a different disturbance after every push, while the ticket is in flight
    if which == 0: ctx.sad_flow(small[0], small[1], 8, 8)
    elif which == 1:
        n = 60 + k % 300
        e = np.zeros((n, 4), np.float32); e[:, 0] = (np.arange(n) % 16 + 0.5) / 16; e[:, 1] = (np.arange(n) // 16 % 9 + 0.5) / 9
        e[:, 2:] = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        ctx.densify(e, 16, 9); ctx.detect(e)
    elif which == 2: ctx.almeida(rot, 16 / 9, 22.275, use_ransac=False)
    elif which == 3: ctx.lk_flow(small[0], small[1], 2, 4, 2)
    elif which == 4: ctx.lk_decode(small[0], small[1], 2, 4, 2)
    elif which == 5: ctx.almeida(rot, 16 / 9, 22.275, use_ransac=True, num_iters=20, num_samples=200, seed=k)
    if prev is not None:
        res = ctx.frame_wait(prev[0])
        check(prev[1], res, outs[prev[1] % 3])
    prev = (t, k)
res = ctx.frame_wait(prev[0]); check(prev[1], res, outs[prev[1] % 3])
print(f"pipeline soak ({'RANSAC' if RANSAC else 'LSQ'} estimator in the stream): {N} frames {W}x{H} b{B} r{R} in {time.perf_counter() - t0:.1f} s with six kinds of other calls in between, "
      f"mismatching frames {bad}")
sys.exit(1 if bad else 0)
