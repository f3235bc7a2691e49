#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the bench line): PCIe-inclusive SAD rate through the
host-pointer entry point, the other BASELINE configs, and the per-stage cost of the device-resident tail
(densify/detect/Almeida) at the sizes the hot path produces.  Run on the GPU box; prints JSON."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ctx = HipContext(0)
    out = {}
    # --- cfg2 through the host-pointer API (H2D of both frames + kernel + D2H of the entries)
    fr = synth.luma_sequence(2, 1920, 1080, 16)
    ms = timeit(lambda: ctx.sad_flow(fr[0], fr[1], 16, 16), n=20)
    out["cfg2_host_api_ms_per_pair"] = round(ms, 4)
    out["cfg2_host_api_Mvectors_per_s"] = round(8040 / ms / 1e3, 2)

    # --- the Decoder::process_frame shape: only the new frame is uploaded (page-locked buffer), previous frame on device
    pin = ctx.pinned_frame(1080, 1920)
    seq = synth.luma_sequence(4, 1920, 1080, 16)
    k = [0]

    def push():
        np.copyto(pin, seq[k[0] % 4]); k[0] += 1
        ctx.push_frame(pin, 16, 16, detector=False, estimator=False, want_entries=True)
    ms = timeit(push, n=40)
    np.copyto(pin, seq[0])
    ms_nocopy = timeit(lambda: ctx.push_frame(pin, 16, 16, detector=False, estimator=False, want_entries=True), n=40)
    out["cfg2_decoder_process_frame"] = {"ms_per_frame_incl_host_copy_into_pinned": round(ms, 4), "ms_per_frame": round(ms_nocopy, 4),
                                         "Mvectors_per_s": round(8040 / ms_nocopy / 1e3, 2),
                                         "note": "2.07 MB H2D from page-locked memory + single-pair search + 129 KB D2H, synchronous"}
    pag = seq[1].copy()
    ms_pageable = timeit(lambda: ctx.push_frame(pag, 16, 16, detector=False, estimator=False, want_entries=True), n=40)
    out["cfg2_decoder_process_frame"]["ms_per_frame_pageable_source"] = round(ms_pageable, 4)
    # single-pair kernel alone (device resident)
    d2 = torch.from_numpy(seq[:2]).cuda()
    o1 = torch.empty((8040, 4), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    out["cfg2_single_pair_kernel_ms"] = round(timeit(lambda: ctx.sad_flow_dev(d2.data_ptr(), 2, 1920, 1080, 1920, 1920 * 1080, 0, 16, 16, o1.data_ptr(), None), n=50), 4)
    # --- device-resident SAD at the other geometries (batched)
    for name, (W, H, B, R, P) in {"cfg1_640x360_b16_r8": (640, 360, 16, 8, 16),
                                  "cfg2_1080p_b16_r16": (1920, 1080, 16, 16, 16),
                                  "cfg4_4k_b8_r32": (3840, 2160, 8, 32, 4),
                                  "1080p_b8_r16": (1920, 1080, 8, 16, 8)}.items():
        frames = synth.luma_sequence(P + 1, W, H, max_step=min(R, 16))
        d = torch.from_numpy(frames).cuda()
        nb = (W // B) * (H // B)
        o = torch.empty((P, nb, 4), dtype=torch.float32, device="cuda")
        ms = timeit(lambda: ctx.sad_flow_dev(d.data_ptr(), P + 1, W, H, W, W * H, 0, B, R, o.data_ptr(), None), n=10)
        out[name] = {"ms_per_pair": round(ms / P, 5), "Mvectors_per_s": round(P * nb / ms / 1e3, 2),
                     "abs_diffs_T_per_s": round(P * nb * B * B * (2 * R + 1) ** 2 / ms / 1e9, 2)}
        if name == "cfg2_1080p_b16_r16":
            ent = o
            P2, nb2 = P, nb

    # --- the tail on device-resident vectors of cfg2 (batch of 16 pairs)
    dim = ctx.block_dim(0.05, 3)
    res = torch.empty((P2, 4), dtype=torch.int32, device="cuda")
    fld = torch.empty((P2, dim * dim, 2), dtype=torch.float32, device="cuda")
    quat = torch.empty((P2, 4), dtype=torch.float32, device="cuda")
    out["tail_cfg2_batch16"] = {
        "densify14_ms": round(timeit(lambda: ctx.densify_dev(ent.data_ptr(), nb2, P2, dim, dim, fld.data_ptr())), 4),
        "detect_ms": round(timeit(lambda: ctx.detect_dev(ent.data_ptr(), nb2, P2, 0.05, 3, 0.003, res.data_ptr(), fld.data_ptr())), 4),
        "almeida_lsq_ms": round(timeit(lambda: ctx.almeida_dev(ent.data_ptr(), nb2, P2, 16 / 9, 22.275, False, 0, 0.05, 0, 0, quat.data_ptr())), 4),
        "almeida_ransac_ms": round(timeit(lambda: ctx.almeida_dev(ent.data_ptr(), nb2, P2, 16 / 9, 22.275, True, 200, 0.05, 1000, 7, quat.data_ptr()), n=5), 4),
    }
    # single-pair latency of the tail (what one Estimator::estimate / Detector::detect_motion call costs)
    out["tail_cfg2_single_pair"] = {
        "detect_ms": round(timeit(lambda: ctx.detect_dev(ent.data_ptr(), nb2, 1, 0.05, 3, 0.003, res.data_ptr(), fld.data_ptr())), 4),
        "almeida_lsq_ms": round(timeit(lambda: ctx.almeida_dev(ent.data_ptr(), nb2, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, quat.data_ptr())), 4),
        "almeida_ransac_ms": round(timeit(lambda: ctx.almeida_dev(ent.data_ptr(), nb2, 1, 16 / 9, 22.275, True, 200, 0.05, 1000, 7, quat.data_ptr()), n=5), 4),
    }
    # --- cfg3: per-pixel 1080p entries (2.07 M records, 33 MB): Almeida LSQ + densify to 150x84
    e = torch.from_numpy(synth.rotation_field(1920, 1080)).cuda()
    n = e.shape[0]
    q1 = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    f150 = torch.empty((150 * 84, 2), dtype=torch.float32, device="cuda")
    ms = timeit(lambda: ctx.almeida_dev(e.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q1.data_ptr()), n=5, warm=1)
    out["cfg3_almeida_lsq_1080p_per_pixel"] = {"ms": round(ms, 3), "Mvectors_per_s": round(n / ms / 1e3, 1),
                                               "GBps_single_pass_bytes": round(16 * n / ms / 1e6, 1)}
    ms = timeit(lambda: ctx.densify_dev(e.data_ptr(), n, 1, 150, 84, f150.data_ptr()), n=5, warm=1)
    ms_r = timeit(lambda: ctx.densify_raster_dev(e.data_ptr(), None, 1920, 1080, 150, 84, f150.data_ptr()), n=5, warm=1)
    out["cfg3_densify_150x84_per_pixel_rectangle_walk"] = {"ms": round(ms_r, 3), "Mvectors_per_s": round(n / ms_r / 1e3, 1),
                                                            "GBps_entry_bytes": round(16 * n / ms_r / 1e6, 1)}
    out["cfg3_densify_150x84_per_pixel"] = {"ms": round(ms, 3), "Mvectors_per_s": round(n / ms / 1e3, 1),
                                            "GBps_entry_bytes": round(16 * n / ms / 1e6, 1)}
    # --- the estimator input cv-decoder really produces in full-resolution mode: <= 150 x 84 down-sampled records
    e84 = torch.from_numpy(synth.rotation_field(150, 84)).cuda()
    ms = timeit(lambda: ctx.almeida_dev(e84.data_ptr(), 150 * 84, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q1.data_ptr()), n=10, warm=2)
    out["almeida_lsq_12600_downsampled_records"] = {"ms": round(ms, 4), "note": "cluster solver (13 workgroups, granule exchange), IEEE division"}
    ms = timeit(lambda: ctx.almeida_dev(e84.data_ptr(), 150 * 84, 1, 16 / 9, 22.275, True, 200, 0.05, 1000, 3, q1.data_ptr()), n=10, warm=2)
    out["almeida_ransac_12600_downsampled_records"] = {"ms": round(ms, 4)}
    # --- cfg3 from pixels: 1080p pair -> 3-level LK flow (r=4, 3 steps/level) -> per-pixel records -> densify -> Almeida
    fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
    dfr = torch.from_numpy(fr).cuda()
    d_ent = torch.empty((1920 * 1080, 4), dtype=torch.float32, device="cuda")
    f84 = torch.empty((150 * 84, 2), dtype=torch.float32, device="cuda")
    ms_lk = timeit(lambda: ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), 1920, 1080, 1920, 3, 4, 3, None, d_ent.data_ptr()), n=5, warm=1)

    def chain():
        ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), 1920, 1080, 1920, 3, 4, 3, None, d_ent.data_ptr())
        ctx.densify_raster_dev(d_ent.data_ptr(), None, 1920, 1080, 150, 84, f84.data_ptr())
        ctx.almeida_dev(d_ent.data_ptr(), 1920 * 1080, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q1.data_ptr())
    ms_chain = timeit(chain, n=5, warm=1)
    out["cfg3_lk_flow_1080p"] = {"ms": round(ms_lk, 3), "Mvectors_per_s": round(1920 * 1080 / ms_lk / 1e3, 1)}
    out["cfg3_chain_lk_densify_almeida"] = {"ms": round(ms_chain, 3), "Mvectors_per_s": round(1920 * 1080 / ms_chain / 1e3, 1)}
    # --- cv-decoder's contrast mask at 1080p (device resident) and the whole hip_lk process_frame through the host API
    g = torch.from_numpy(synth.flatten_regions(fr[1:2], region=96, seed=3)[0]).cuda()
    d_mask = torch.empty((1080, 1920), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: ctx.contrast_mask_dev(g.data_ptr(), 1920, 1080, 1920, d_mask.data_ptr()), n=20)
    out["contrast_mask_1080p"] = {"ms": round(ms, 4), "GBps_algorithmic_2B_per_px": round(2 * 1920 * 1080 / ms / 1e6, 1)}
    ctx.use_own_stream()
    frm = synth.flatten_regions(fr, region=96, seed=3)
    ms = timeit(lambda: ctx.lk_decode(frm[0], frm[1], contrast_mask=True), n=5, warm=1)
    out["hip_lk_process_frame_1080p_host_api_masked_ms"] = round(ms, 3)
    ms = timeit(lambda: ctx.lk_decode(frm[0], frm[1]), n=5, warm=1)
    out["hip_lk_process_frame_1080p_host_api_unmasked_ms"] = round(ms, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
