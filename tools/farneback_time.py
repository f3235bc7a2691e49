#!/usr/bin/env python3
"""Time of one 1080p Farneback pair on the device (HIP events), per kernel with rocprofv3 when wrapped:
  python tools/farneback_time.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, H = 1920, 1080
ctx = HipContext(0)
ctx.use_torch_stream()
fr = synth.luma_sequence(2, W, H, max_step=3, seed=11)
d = torch.from_numpy(fr).cuda()
d_ent = torch.empty((W * H, 4), dtype=torch.float32, device="cuda")


def step():
    ctx.farneback_flow_dev(d[0].data_ptr(), d[1].data_ptr(), W, H, W, levels=levels, d_out_entries=d_ent.data_ptr())
for _ in range(5):
    step()
ctx.sync(); ctx.timer_start()
for _ in range(reps):
    step()
ms = ctx.timer_stop() / reps
px = sum(((W + (1 << k) // 2) >> k) * ((H + (1 << k) // 2) >> k) for k in range(levels + 1))
# algorithmic HBM bytes per pair -- the formula of bench_legs.py's roofline_farneback: per layer pixel 40 (expansion planes
# written) + 60 (first matrices: R0 + R1 read, M written) + 2 x 80 (updates 1, 2: M + R0 + R1 read, next M written) + 28 (last update: M
# read, flow written) = 288 B; + 16 B of records and 2 B of frames per frame pixel
algo = px * 288 + W * H * 18
print(json.dumps({"levels": levels, "farneback_1080p_ms": round(ms, 4), "algorithmic_MB": round(algo / 1e6, 1), "GBs_at_algorithmic_bytes": round(algo / (ms * 1e-3) / 1e9, 1)}))
ctx.use_own_stream(); ctx.close()
