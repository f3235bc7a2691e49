#!/usr/bin/env python3
"""Launch time of the strip kernel for every (block, range) it is instantiated for, 1080p, 32 resident pairs (HIP events).
  python tools/sad_geometry_time.py [--lib <libofps_hip.so>]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from ofps_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import json  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402

W, H, P = 1920, 1080, 32
fr = synth.luma_sequence(P + 1, W, H, max_step=8)
ctx = HipContext(0)
ctx.use_torch_stream()
d = torch.from_numpy(np.ascontiguousarray(fr)).cuda()
out = {}
for B in (16, 8):
    for R in (8, 12, 16, 20, 24, 28, 32):
        nblk = (W // B) * (H // B)
        o = torch.empty((P, nblk, 4), dtype=torch.float32, device="cuda")

        def step():
            ctx.sad_flow_dev(d.data_ptr(), P + 1, W, H, W, W * H, 0, B, R, o.data_ptr(), None)
        for _ in range(3):
            step()
        ctx.sync(); ctx.timer_start()
        for _ in range(10):
            step()
        ms = ctx.timer_stop() / 10
        absd = P * nblk * B * B * (2 * R + 1) ** 2
        out[f"{B}x{B}+-{R}"] = {"ms": round(ms, 4), "sad_unit_frac": round(absd / (ms * 1e-3) / (256 * 4 * 2.4e9 * 64), 4)}
print(json.dumps(out))
