#!/usr/bin/env python3
"""Wall time of ofps_hip_lk_decode (hip_lk's Decoder::process_frame shape: two host frames in, down-sampled records out)
at 1080p, with and without cv-decoder's contrast mask."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0)
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
