#!/usr/bin/env python3
"""Writes tools/external_parity/data: three seeded .mvec clips and, per frame, what this build says the reference's
estimator / detector / densifier return on them (README.md).  Default: everything from the CPU oracle; --hip: everything
from libofps_hip.so (needs a GPU)."""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from ofps_amd import mvec, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hip", action="store_true", help="take every number from the HIP path instead of the oracle")
    args = ap.parse_args()
    out = os.path.join(HERE, "data")
    os.makedirs(out, exist_ok=True)
    W, H, B, R, F = 640, 360, 16, 8, 6
    aspect, fov = 16 / 9, 39.6 * 9 / 16
    if args.hip:
        from ofps_amd.runtime import HipContext
        ctx = HipContext(0)
        sad = lambda a, b: ctx.sad_flow(a, b, B, R)
        lsq = lambda e: ctx.almeida(e, aspect, fov, use_ransac=False)[0]
        det = lambda e: ctx.detect(e)
        dens = lambda e, w, h: ctx.densify(e, w, h, want_cells=True)
    else:
        import oracle
        cam = oracle.camera(aspect, fov)
        sad = lambda a, b: oracle.sad_flow(a, b, B, R)[0]
        lsq = lambda e: oracle.solve_ypr_given(e, cam)
        det = lambda e: oracle.detect_motion(e)
        dens = lambda e, w, h: oracle.densify(e, w, h, want_cells=True)
    for clip in range(3):
        fr = synth.luma_sequence(F, W, H, max_step=R, seed=synth.SEED0 + 500 + clip)
        frames = [np.zeros((0, 4), np.float32)]                    # first frame of a stream: no vectors (count 0)
        for k in range(1, F):
            e = np.array(sad(fr[k - 1], fr[k]), np.float32)
            if clip == 1:
                e[:, 2:] *= np.float32(0.02)                        # a nearly static clip: detector says None
            if clip == 2 and k == 3:                                # positions the clamp collapses
                e[0, :2] = (-0.25, 0.5); e[1, :2] = (0.5, 1.5); e[2, :2] = (0.0, 0.3); e[3, :2] = (1.0, 0.7)
                e[4, :2] = (np.nan, 0.5); e[5, :2] = (0.5, np.inf); e[6, :2] = (-1e-9, 1.0)
            frames.append(e)
        with open(os.path.join(out, f"clip{clip}.mvec"), "wb") as f:
            for e in frames:
                mvec.write_frame(f, e)
        exp = []
        for k, e in enumerate(frames):
            d = det(e) if len(e) else None
            f14, cells = dens(e, 14, 14) if len(e) else (np.zeros((14, 14, 2), np.float32), np.zeros((0, 2), np.uint32))
            f150, _ = dens(e, 60, 34) if len(e) else (np.zeros((34, 60, 2), np.float32), None)
            q = lsq(e) if len(e) else np.array([1, 0, 0, 0], np.float32)
            exp.append({"clip": clip, "frame": k, "quat": [float(x) for x in q],
                        "detect_area": None if d is None else int(d[0]),
                        "detect_field": [] if d is None else [int(x) for x in np.asarray(d[1], np.float32).view(np.uint32).ravel()],
                        "cells": [[int(c[0]), int(c[1])] for c in np.asarray(cells)],
                        "field_14x14": [int(x) for x in np.asarray(f14, np.float32).view(np.uint32).ravel()],
                        "field_60x34": [int(x) for x in np.asarray(f150, np.float32).view(np.uint32).ravel()]})
        with open(os.path.join(out, f"expected_clip{clip}.json"), "w") as f:
            json.dump({"source": "hip" if args.hip else "oracle", "geometry": [W, H, B, R], "camera": [aspect, fov], "frames": exp}, f)
        print(f"clip{clip}: {F} frames, {sum(len(e) for e in frames)} vectors")


if __name__ == "__main__":
    main()
