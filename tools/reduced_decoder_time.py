#!/usr/bin/env python3
"""The reduced ("Process Fullres" = false) dense decoders from the host: 1080p BGR / luma frames from page-locked memory, read-ahead form
(two tickets), resize -> gray -> mask + flow on 150 x 84 -> one record per unmasked reduced-frame pixel; hip_flow (cv-decoder's flow, previous
flow carried over) and hip_lk.  Also pageable frames (upload path instead of the gather from host memory).
  python tools/reduced_decoder_time.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W, H = 1920, 1080
ctx = HipContext(0)
y = synth.luma_sequence(4, W, H, max_step=3, seed=11)
rng = np.random.default_rng(11)
bgr = np.clip(y[..., None].astype(int) + rng.integers(-40, 41, (1, H, W, 3)), 0, 255).astype(np.uint8)
pins_bgr = [ctx.pinned_frame(H, 3 * W).reshape(H, W, 3) for _ in range(4)]
pins_y = [ctx.pinned_frame(H, W) for _ in range(4)]
for k in range(4):
    np.copyto(pins_bgr[k], bgr[k]); np.copyto(pins_y[k], y[k])
out = [np.zeros((150 * 150, 4), np.float32) for _ in range(2)]


def run(frames, n, **kw):
    prev = None
    for k in range(n):
        t = ctx.lk_push_frame_async(frames[k % 4], **kw)
        if prev is not None:
            ctx.lk_frame_wait(prev, out[k & 1])
        prev = t
    return ctx.lk_frame_wait(prev, out[n & 1])


FLOW = dict(levels=5, radius=6, iters=3, farneback=True, use_previous=True, contrast_mask=True, reduced=True)
LK = dict(levels=3, radius=4, iters=3, contrast_mask=True, reduced=True)
for name, frames, kw in (("hip_flow, BGR page-locked", pins_bgr, dict(FLOW, fmt=ctx.FMT_BGR)), ("hip_flow, luma page-locked", pins_y, dict(FLOW, fmt=ctx.FMT_LUMA)),
                         ("hip_flow, BGR pageable", list(bgr), dict(FLOW, fmt=ctx.FMT_BGR)), ("hip_lk, BGR page-locked", pins_bgr, dict(LK, fmt=ctx.FMT_BGR)),
                         ("hip_flow, BGR page-locked, full resolution (the default mode)", pins_bgr, dict(FLOW, fmt=ctx.FMT_BGR, reduced=False))):
    ctx.lk_reset(); run(frames, 8, **kw)
    vals = []
    for _ in range(3):
        t0 = time.perf_counter(); r = run(frames, N, **kw); vals.append((time.perf_counter() - t0) / N * 1e3)
    print(f"{name}: {sorted(vals)[1]:.4f} ms per 1080p frame (min {min(vals):.4f}, max {max(vals):.4f}; {len(r[0])} records on {r[1][0]} x {r[1][1]})")
ctx.lk_reset(); ctx.close()
