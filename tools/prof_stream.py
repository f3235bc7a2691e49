#!/usr/bin/env python3
"""One `hip_sad` + `hip_block_motion` + `hip_almeida` frame loop (ofps_hip_push_frame) over a short 1080p sequence;
meant to be wrapped in `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown of the cfg5 path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    ctx = HipContext(0)
    fr = synth.luma_sequence(8, 1920, 1080, max_step=8, seed=3)
    pin = ctx.pinned_frame(1080, 1920)
    for k in range(n):
        pin[:] = fr[k % 8]
        ctx.push_frame(pin, 16, 16, aspect=16 / 9, fov_y_deg=22.275, use_ransac=bool(k & 1))
    ctx.close()


if __name__ == "__main__":
    main()
