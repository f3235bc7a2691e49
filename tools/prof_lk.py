#!/usr/bin/env python3
"""Runs the cfg3 chain (1080p pair -> 3-level LK flow -> records -> [mask/compact] -> densify -> Almeida) a few times;
meant to be wrapped in `rocprofv3 --kernel-trace --stats` for a per-kernel breakdown."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    max_step = int(sys.argv[2]) if len(sys.argv) > 2 else 3      # 3: the +-3 px content; 16: the SAD bench's +-16 px regions
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # Gauss-Newton steps per level (instruction split: tools/lk_instr_split.sh)
    ctx = HipContext(0)
    ctx.use_torch_stream()
    fr = synth.luma_sequence(2, 1920, 1080, max_step=max_step, seed=11)
    dfr = torch.from_numpy(fr).cuda()
    d_ent = torch.empty((1920 * 1080, 4), dtype=torch.float32, device="cuda")
    f84 = torch.empty((150 * 84, 2), dtype=torch.float32, device="cuda")
    q1 = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    for _ in range(n):
        ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), 1920, 1080, 1920, 3, 4, iters, None, d_ent.data_ptr())
        ctx.densify_raster_dev(d_ent.data_ptr(), None, 1920, 1080, 150, 84, f84.data_ptr())
        ctx.almeida_dev(d_ent.data_ptr(), 1920 * 1080, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q1.data_ptr())
    torch.cuda.synchronize()
    helped = ctx.lk_helped_tiles()                     # tiles a waiting child computed itself: extra work in the profile (0 on a whole device)
    ctx.use_own_stream()
    ctx.close()
    if helped:
        print(f"prof_lk: {helped} tiles of the LK pyramid were computed by a waiting child -- the profile contains duplicate work", file=sys.stderr)


if __name__ == "__main__":
    main()
