#!/usr/bin/env python3
"""Hostile arguments through the C ABI (host-pointer entry points): every call is a valid small call with ONE argument replaced by an
invalid value -- NULL pointers, zero / negative / INT_MIN sizes, a stride below the width, unknown flags, out-of-range levels /
radius / iterations / block / range, NaN and negative floats, an absent ticket.  The library must answer with an error code or a
harmless success -- never crash, hang or corrupt the context: after all of them the context still gives a fresh context's bits.
usage: badarg_fuzz.py"""
import ctypes as C
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth, _lib
from ofps_amd.runtime import HipContext

ctx = HipContext(0)
L, h = ctx._lib, ctx._h
W, H = 64, 48
fr = synth.luma_sequence(3, W, H, max_step=2, seed=3)
u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
f32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
big = np.zeros(1 << 18, np.float32); big2 = np.zeros(1 << 18, np.float32); ibuf = np.zeros(1 << 16, np.int32); mask = np.zeros(W * H, np.uint8)
n_out = C.c_size_t(0); gw = C.c_int(0); gh = C.c_int(0); have = C.c_int(0); tick = C.c_int(0)
ent = np.zeros((200, 4), np.float32); ent[:, :2] = np.random.default_rng(1).uniform(0, 1, (200, 2)); ent[:, 2:] = 0.004
prm = _lib.FrameParams(16, 8, 1, 0.05, 3, 0.003, 1, 16 / 9, 22.275, 0, 20, 0.05, 100, 0)
res = (_lib.FrameResult * 4)()
INT_BAD = [0, -1, -2 ** 31]
NULL = None

# (name, argument list, {argument index: [invalid values]})
calls = [
 ("ofps_hip_sad_flow", [h, u8(fr[0]), u8(fr[1]), W, H, W, 16, 8, f32(big), ibuf.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_out)],
  {1: [NULL], 2: [NULL], 3: INT_BAD, 4: INT_BAD, 5: INT_BAD + [W - 1], 6: INT_BAD + [3, 12, 5], 7: INT_BAD + [3, 1000], 8: [NULL]}),
 ("ofps_hip_lk_flow", [h, u8(fr[0]), u8(fr[1]), W, H, W, 2, 4, 2, f32(big), f32(big2)],
  {1: [NULL], 2: [NULL], 3: INT_BAD, 4: INT_BAD, 5: INT_BAD + [W - 1], 6: INT_BAD + [9], 7: INT_BAD + [16], 8: INT_BAD + [65]}),
 ("ofps_hip_lk_decode", [h, u8(fr[0]), u8(fr[1]), W, H, W, 2, 4, 2, 40, 40, 0, f32(big), C.byref(n_out), C.byref(gw), C.byref(gh)],
  {1: [NULL], 2: [NULL], 3: INT_BAD, 4: INT_BAD, 5: [W - 1, -1], 6: [0, 9], 7: [0, 16], 8: [0, 65], 9: INT_BAD, 10: INT_BAD, 11: [0xFF, 4], 12: [NULL], 13: [NULL]}),
 ("ofps_hip_lk_push_frame", [h, u8(fr[0]), W, H, W, 2, 4, 2, 40, 40, 0, f32(big), C.byref(n_out), C.byref(gw), C.byref(gh), C.byref(have)],
  {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: [W - 1], 5: [0, 9], 6: [0, 16], 7: [0, 65], 8: INT_BAD, 9: INT_BAD, 10: [0xFF], 11: [NULL], 12: [NULL], 15: [NULL]}),
 ("ofps_hip_lk_push_frame_async", [h, u8(fr[0]), W, H, W, 2, 4, 2, 40, 40, 0, C.byref(tick)],
  {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: [W - 1], 5: [0, 9], 10: [0xFF], 11: [NULL]}),
 ("ofps_hip_lk_frame_wait", [h, 12345, f32(big), C.byref(n_out), C.byref(gw), C.byref(gh), C.byref(have)], {1: [12345, -1]}),
 ("ofps_hip_contrast_mask", [h, u8(fr[0]), W, H, W, u8(mask)], {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: [W - 1, -1], 5: [NULL]}),
 ("ofps_hip_densify", [h, f32(ent), 200, 16, 9, f32(big), None], {1: [NULL], 3: INT_BAD + [70000], 4: INT_BAD + [70000], 5: [NULL]}),
 ("ofps_hip_densify_interpolated", [h, f32(ent), 200, 16, 9, f32(big)], {1: [NULL], 3: INT_BAD, 4: INT_BAD, 5: [NULL]}),
 ("ofps_hip_densify_to_entries", [h, f32(ent), 200, 16, 9, f32(big), C.byref(n_out)], {1: [NULL], 3: INT_BAD, 4: INT_BAD, 5: [NULL], 6: [NULL]}),
 ("ofps_hip_detect", [h, f32(ent), 200, 0.05, 3, 0.003, C.byref(have), C.byref(n_out), C.byref(gw), f32(big)],
  {1: [NULL], 3: [0.0, -1.0, float("nan"), 5.0], 4: [0], 5: [float("nan"), -1.0], 6: [NULL]}),
 ("ofps_hip_almeida", [h, f32(ent), 200, 16 / 9, 22.275, 0, 20, 0.05, 100, 0, f32(big), None],
  {1: [NULL], 3: [0.0, -1.0, float("nan")], 4: [0.0, -5.0, 180.0, float("nan")], 10: [NULL]}),
 ("ofps_hip_almeida", [h, f32(ent), 200, 16 / 9, 22.275, 1, 20, 0.05, 100, 0, f32(big), None],
  {6: [0], 7: [float("nan"), -1.0], 8: [0, 1, 2]}),
 ("ofps_hip_stage_frame", [h, u8(fr[0]), W, H, W], {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: [W - 1]}),
 ("ofps_hip_push_frame", [h, u8(fr[1]), W, H, W, C.byref(prm), C.byref(res[0]), f32(big), f32(big2)],
  {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: [W - 1], 5: [NULL], 6: [NULL]}),
 ("ofps_hip_push_frame_async", [h, u8(fr[1]), W, H, W, C.byref(prm), f32(big), f32(big2), C.byref(tick)],
  {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: [W - 1], 5: [NULL], 8: [NULL]}),
 ("ofps_hip_push_frames_async", [h, u8(fr), 3, W, H, W, W * H, C.byref(prm), f32(big), C.byref(tick)],
  {1: [NULL], 2: INT_BAD, 3: INT_BAD, 4: INT_BAD, 5: [W - 1], 6: [0, W * H - 1], 7: [NULL], 9: [NULL]}),
 ("ofps_hip_frame_wait", [h, 777, C.byref(res[0])], {1: [777, -1], 2: [NULL]}),
 ("ofps_hip_frames_wait", [h, 777, res], {1: [777, -1], 2: [NULL]}),
 ("ofps_hip_set_option", [h, b"OFPS_HIP_ALMEIDA_EPT", b"3"], {1: [NULL, b"NO_SUCH"], 2: [b"3", b"x"]}),
 ("ofps_hip_set_sad_mode", [h, 0], {1: [-1, 99]}),
]
bad_params = []                                     # the per-frame parameter block, field by field
for fld, vals in {"block": [0, 3, -16], "range": [0, 3, -8, 1000], "min_size": [0.0, float("nan")], "subdivide": [0], "aspect": [0.0, float("nan")],
                  "fov_y_deg": [0.0, 180.0, float("nan")]}.items():
    for v in vals:
        p = _lib.FrameParams(16, 8, 1, 0.05, 3, 0.003, 1, 16 / 9, 22.275, 0, 20, 0.05, 100, 0); setattr(p, fld, v); bad_params.append((fld, v, p))

n_calls = n_err = 0
accepted = []
def drain():
    """collect whatever a mutated asynchronous call may have enqueued, so later calls see a quiet context"""
    for t in range(0, 4):
        L.ofps_hip_frame_wait(h, t, C.byref(res[0])); L.ofps_hip_frames_wait(h, t, res)
        L.ofps_hip_lk_frame_wait(h, t, f32(big), C.byref(n_out), C.byref(gw), C.byref(gh), C.byref(have))
    L.ofps_hip_reset_frames(h); L.ofps_hip_lk_reset(h)
good_prm = _lib.FrameParams(16, 8, 1, 0.05, 3, 0.003, 1, 16 / 9, 22.275, 0, 20, 0.05, 100, 0)
def prime(name):
    """stream entry points: a valid first frame in front, so that the mutated call is the one that has work to do"""
    if "lk_push_frame" in name:
        L.ofps_hip_lk_reset(h)
        L.ofps_hip_lk_push_frame(h, u8(fr[2]), W, H, W, 2, 4, 2, 40, 40, 0, f32(big), C.byref(n_out), C.byref(gw), C.byref(gh), C.byref(have))
    elif name in ("ofps_hip_push_frame", "ofps_hip_push_frame_async", "ofps_hip_push_frames_async"):
        L.ofps_hip_reset_frames(h)
        L.ofps_hip_push_frame(h, u8(fr[0]), W, H, W, C.byref(good_prm), C.byref(res[0]), f32(big), f32(big2))
for name, args, muts in calls:
    fn = getattr(L, name)
    for idx, vals in muts.items():
        for v in vals:
            prime(name)
            a = list(args); a[idx] = v
            rc = fn(*a); n_calls += 1; n_err += rc != 0
            if rc == 0: accepted.append(f"{name}[{idx}]={v!r}")
            if name.endswith("_async") or name.endswith("push_frame"): drain()
for fld, v, p in bad_params:
    prime("ofps_hip_push_frame")
    rc = L.ofps_hip_push_frame(h, u8(fr[1]), W, H, W, C.byref(p), C.byref(res[0]), f32(big), f32(big2)); n_calls += 1; n_err += rc != 0
    if rc == 0: accepted.append(f"push_frame.params.{fld}={v!r}")
    drain()
# the context still works, and gives a fresh context's bits
clean = HipContext(0)
ok = np.array_equal(ctx.sad_flow(fr[0], fr[1], 16, 8).view(np.uint32), clean.sad_flow(fr[0], fr[1], 16, 8).view(np.uint32))
ok &= np.array_equal(ctx.lk_flow(fr[0], fr[1], 2, 4, 2).view(np.uint32), clean.lk_flow(fr[0], fr[1], 2, 4, 2).view(np.uint32))
a, b = ctx.lk_decode(fr[0], fr[1], 2, 4, 2), clean.lk_decode(fr[0], fr[1], 2, 4, 2)
ok &= np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
ctx.reset_frames(); clean.reset_frames(); ctx.push_frame(fr[0]); clean.push_frame(fr[0])
a, b = ctx.push_frame(fr[1], want_entries=True), clean.push_frame(fr[1], want_entries=True)
ok &= np.array_equal(a["entries"].view(np.uint32), b["entries"].view(np.uint32)) and np.array_equal(a["quat"].view(np.uint32), b["quat"].view(np.uint32))
print(f"bad-argument fuzz: {n_calls} calls with one invalid argument each, {n_err} answered with an error code, {n_calls - n_err} with a harmless success; "
      f"the context afterwards gives a fresh context's bits: {bool(ok)}")
if "-v" in sys.argv: print("accepted:", "; ".join(accepted))
sys.exit(0 if ok else 1)
