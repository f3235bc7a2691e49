#!/usr/bin/env python3
"""(Re)writes the EXPECTED outputs in data/farneback_pairs.npz from the build's CPU restatement (oracle/farneback_oracle.c; run in the build's
container; the two committed frame pairs themselves are left as they are).  For each pair ("camera", "regions"):
    <n>_flow, <n>_flow_warm                     cv-decoder's call (cv-decoder/src/lib.rs:188-199), cold and started from the cold flow (:161-165)
    <n>_flow_v1, <n>_flow_v2                    the same with the other published forms of OpenCV's separable Gaussian (ascending row taps
                                                beyond 5 taps; + fused multiply-adds): opencv_compare.py reports which one a given build runs
    <n>_stage_levels0_iters1, _levels0, _levels1, _levels2   ablations that localise a difference: one layer / one update (expansion + one
                                                solve, no pyramid), one layer (updates), two and three layers (blur + resize enter)
    <n>_layer<k>                                the layer images I_k, k = 1 .. 3 (GaussianBlur of the float frame + resize INTER_LINEAR)
    <n>_layer<k>_v1                             ... with ascending row taps (differs from k = 2 on: 9 taps and more)
    <n>_gray_small, <n>_bgr, <n>_bgr_small_gray the front-end (cv-decoder/src/lib.rs:124-135): the frame resized to 150 x 84; a BGR version of
                                                the frame; that BGR frame resized then converted
tests/test_farneback_oracle.py keeps every array equal to what the oracle computes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

PATH = os.path.join(ROOT, "tools", "external_parity", "data", "farneback_pairs.npz")


def expected(prev, cur, name):
    out = {}
    cold = oracle.farneback_flow(prev, cur)
    out[name + "_flow"] = cold
    out[name + "_flow_warm"] = oracle.farneback_flow(prev, cur, init=cold)
    for v in (1, 2):
        with oracle.farneback_blur_variant(v):
            out[f"{name}_flow_v{v}"] = oracle.farneback_flow(prev, cur)
    out[name + "_stage_levels0_iters1"] = oracle.farneback_flow(prev, cur, levels=0, iters=1)
    out[name + "_stage_levels0"] = oracle.farneback_flow(prev, cur, levels=0)
    out[name + "_stage_levels1"] = oracle.farneback_flow(prev, cur, levels=1)
    out[name + "_stage_levels2"] = oracle.farneback_flow(prev, cur, levels=2)
    for k in (1, 2, 3):
        if k < len(oracle.farneback_layers(prev.shape[1], prev.shape[0], 5)):
            out[f"{name}_layer{k}"] = oracle.farneback_layer(cur, k)[0]
            with oracle.farneback_blur_variant(1):
                out[f"{name}_layer{k}_v1"] = oracle.farneback_layer(cur, k)[0]
    H, W = cur.shape
    rng = np.random.default_rng(7)
    bgr = np.clip(cur[..., None].astype(int) + rng.integers(-40, 41, (H, W, 3)), 0, 255).astype(np.uint8)
    gw, gh = oracle.cv_grid(W, H)
    out[name + "_gray_small"] = oracle.resize_linear(cur, gw, gh)
    out[name + "_bgr"] = bgr
    out[name + "_bgr_small_gray"] = oracle.cv_frontend(bgr, oracle.FMT_BGR, process_fullres=False)
    return out


def main():
    d = dict(np.load(PATH))
    new = {}
    for name in ("camera", "regions"):
        new[name + "_prev"], new[name + "_cur"] = d[name + "_prev"], d[name + "_cur"]
        new.update(expected(d[name + "_prev"], d[name + "_cur"], name))
        assert np.array_equal(new[name + "_flow"].view(np.uint32), d[name + "_flow"].view(np.uint32)), "the committed cold-start flow is no longer the oracle's"
    np.savez_compressed(PATH, **new)
    print({k: v.shape for k, v in new.items()})


if __name__ == "__main__":
    main()
