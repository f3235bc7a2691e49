#!/usr/bin/env python3
"""Bounded legs of bench.py for the BASELINE.json configs the headline line is NOT quoted on (N = 1 only; never `value`):

  cfg3_chain   configs[2]: 1080p pair -> 3-level pyramidal LK (hip_lk) -> per-pixel records -> densify 150x84 -> Almeida
               LSQ on 2,073,600 records; per-stage HIP-event times, chain wall time, rooflines of the two dominant kernels
               (LK level kernel: f32 VALU operations of the spec; cluster solver: record bytes read once vs HBM);
  cfg4         configs[3] at one GPU: 4K, 8x8 blocks, +-32, one resident batch of 64 pairs; Mvectors/s and the SAD-unit
               fraction;
  cfg5_stream  configs[4]: 1080p@60 over loop-back TCP, SAD + block-motion + Almeida fused per frame
               (ofps_hip_push_frame); frame received -> island + quaternion on the host, p50 / p99.

Every leg runs its own oracle spot-check AFTER its timed region (the oracle is the checker, never the thing measured).
bench.py starts this file in a fresh process (`python bench_legs.py`), so the legs see the HIP runtime state of a plugin
host, and merges the JSON object it prints into the bench line.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
# f32 vector peak of MI355X_MICROARCH.md: 157.3 TFLOP/s = 256 CU x 4 SIMD x 32 lanes per clock x 2.4 GHz x 2 (a fused
# multiply-add counts as two flops).  The same machine in INSTRUCTIONS: 78.6 T lane-instructions/s (tools/ubench_valu measures 2.5
# cycles per wave64 instruction, i.e. 25.6 lanes per clock: the nominal figure is the stricter peak).  One unit on both sides
# of each fraction (round 3 divided fma = 2 operations by the instruction rate).
VALU_F32_PEAK_FLOPS = 256 * 4 * 32 * 2.4e9 * 2
VALU_F32_PEAK_INSTR = 256 * 4 * 32 * 2.4e9
SAD_PEAK = 256 * 4 * 2.4e9 * 64.0                          # |a-b| per second, see bench.py
LK_SPEC_FLOPS_PER_TAP = 11                                 # DESIGN.md N2, spec revision 2: per tap and step 3 subtractions + 3 lerp fma + 2 residual fma
                                                           # = 8 instructions' worth of arithmetic of which the kernel issues 7 (the horizontal lerp of a
                                                           # sample row serves two window rows); flops with fma = 2: 11
LK_SPEC_INSTR_PER_TAP = 7


class QuietGC:
    """CPython's cyclic collector inside a host-timed loop is an artefact of the measuring process, not of the path measured: after
    `import torch` a generation-2 collection walks ~170,000 objects and stops the thread for 35-50 ms -- once, at an allocation
    count that lands in whichever loop happens to be running (round 4: 40 ms inside the 300-frame read-ahead loop = +0.13 ms per
    frame; round 3: the same pause inside the host-copy loop; profiles/r05/read_ahead_bisect.txt).  Inside the block everything
    alive is moved to the permanent generation (gc.freeze, what a long-running Python host does after start-up); the young
    generations keep running, and every collection that still happens inside the block is counted and timed, so a number
    measured here says what the collector cost it."""

    def __enter__(self):
        import gc
        self._gc = gc
        gc.collect()
        gc.freeze()
        self.events = []                                      # [generation, ms]
        self._t = 0.0

        def cb(phase, info):
            if phase == "start":
                self._t = time.perf_counter()
            else:
                self.events.append([info["generation"], round((time.perf_counter() - self._t) * 1e3, 3)])
        self._cb = cb
        gc.callbacks.append(cb)
        return self

    def __exit__(self, *exc):
        self._gc.callbacks.remove(self._cb)
        self._gc.unfreeze()
        return False

    def summary(self):
        return {"collections": len(self.events), "oldest_generation": max((e[0] for e in self.events), default=None),
                "total_ms": round(sum(e[1] for e in self.events), 3), "longest_ms": max((e[1] for e in self.events), default=0.0)}


def median_min_max(values, digits=4):
    v = sorted(values)
    return {"median": round(v[len(v) // 2], digits), "min": round(v[0], digits), "max": round(v[-1], digits), "repeats": len(v)}


def _event_ms(ctx, fn, reps, warm=25, groups=5):
    """Milliseconds of fn() between HIP events on the context's stream: the MEDIAN of `groups` averages over reps / groups back-to-back
    calls each (round 6: the perf gate compares medians; one average moved +-4 % from collection to collection)."""
    for _ in range(warm):
        fn()
    per = max(1, reps // groups)
    vals = []
    for _ in range(groups):
        ctx.sync()
        ctx.timer_start()
        for _ in range(per):
            fn()
        vals.append(ctx.timer_stop() / per)
    vals.sort()
    return vals[len(vals) // 2]


def cfg3_chain_leg(device: int = 0, reps: int = 60) -> dict:
    import torch
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    W, H, LV, RAD, IT, GW, GH = 1920, 1080, 3, 4, 3, 150, 84
    ctx = HipContext(device)
    ctx.use_torch_stream()
    n = W * H
    d_ent = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_fld = torch.empty((GW * GH, 2), dtype=torch.float32, device="cuda")
    d_q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    # two contents: +-3 px motion, and the +-16 px region motion the SAD bench line itself uses, whose
    # region borders give tiles with incoherent flows (the grouped path of the level kernel)
    contents = {"pm3": synth.luma_sequence(2, W, H, max_step=3, seed=11), "pm16": synth.luma_sequence(2, W, H, max_step=16, seed=11)}
    px = sum((W >> l) * (H >> l) for l in range(LV))
    taps = px * IT * (2 * RAD + 1) ** 2
    per = {}
    kept = {}
    waits0 = ctx.lk_helped_tiles()
    for name, fr in contents.items():
        dfr = torch.from_numpy(fr).cuda()

        def lk():
            ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), W, H, W, LV, RAD, IT, None, d_ent.data_ptr())

        def den():
            ctx.densify_raster_dev(d_ent.data_ptr(), None, W, H, GW, GH, d_fld.data_ptr())

        def alm():
            ctx.almeida_dev(d_ent.data_ptr(), n, 1, W / H, 39.6 * H / W, False, 0, 0.05, 0, 0, d_q.data_ptr())

        def chain():
            lk(); den(); alm()

        def fb():            # the same stage in the reference's own algorithm family: Farneback with cv-decoder's arguments (hip_flow)
            ctx.farneback_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), W, H, W, d_out_entries=d_ent.data_ptr())
        fb_ms = _event_ms(ctx, fb, reps // 2, warm=5)
        kept_fb = d_ent.cpu().numpy() if name == "pm3" else None
        lk_ms = _event_ms(ctx, lk, reps)
        den_ms = _event_ms(ctx, den, reps)
        alm_ms = _event_ms(ctx, alm, reps)
        for _ in range(10):
            chain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            chain()
        torch.cuda.synchronize()
        chain_ms = (time.perf_counter() - t0) / reps * 1e3
        if kept_fb is not None:
            kept["farneback_pm3"] = kept_fb
        per[name] = {"lk_ms": round(lk_ms, 4), "farneback_ms": round(fb_ms, 4), "densify_ms": round(den_ms, 4), "almeida_ms": round(alm_ms, 4),
                     "chain_ms": round(chain_ms, 4),
                     "Mvectors_per_s_chain": round(n / chain_ms / 1e3, 1),
                     "lk_frac_of_f32_peak": round(taps * LK_SPEC_FLOPS_PER_TAP / (lk_ms * 1e-3) / VALU_F32_PEAK_FLOPS, 4)}
        kept[name] = (d_ent.cpu().numpy(), d_q.cpu().numpy()[0], d_fld.cpu().numpy())
        del dfr
    # tiles of the one-launch pyramid that a waiting child computed itself (include/ofps_hip.h): duplicate work inside the timed region -- 0 on a
    # whole device with an in-order dispatcher; reported with the times, the flows are the same bits either way
    lk_waits = ctx.lk_helped_tiles() - waits0
    # the two dense-flow DECODERS as a host drives them (cv-decoder's process_frame shape): frames from page-locked memory one by one,
    # two tickets in flight, cv-decoder's contrast mask + 150 x 84 down-sampling, the frame's records back on the host.  hip_flow keeps
    # the previous frame's pyramid + polynomial expansion on the device (ofps_hip_flow_cache_hits) and starts every pair after the first from
    # the previous pair's flow, as cv-decoder does (OFPS_HIP_FLOW_USE_PREVIOUS; cv-decoder/src/lib.rs:161-165).
    fr4 = synth.luma_sequence(4, W, H, max_step=3, seed=11)
    pins = [ctx.pinned_frame(H, W) for _ in range(4)]
    for k in range(4):
        np.copyto(pins[k], fr4[k])
    outs = [np.zeros((150 * 150, 4), np.float32) for _ in range(2)]

    def stream(nfr, **kw):
        prev = None
        for k in range(nfr):
            t = ctx.lk_push_frame_async(pins[k % 4], contrast_mask=True, **kw)
            if prev is not None:
                ctx.lk_frame_wait(prev, outs[k & 1])
            prev = t
        ctx.lk_frame_wait(prev, outs[nfr & 1])
    decoders = {}
    for name, kw in (("hip_lk", dict(levels=LV, radius=RAD, iters=IT)), ("hip_flow", dict(levels=5, radius=6, iters=3, farneback=True, use_previous=True))):
        ctx.lk_reset()
        stream(8, **kw)
        h0 = ctx.flow_cache_hits()
        with QuietGC():
            runs = []
            for _ in range(5):
                t0 = time.perf_counter()
                stream(100, **kw)
                runs.append((time.perf_counter() - t0) / 100 * 1e3)
        mm = median_min_max(runs)
        decoders[name] = {"ms_per_frame": mm["median"], "ms_per_frame_min": mm["min"], "ms_per_frame_max": mm["max"], "repeats": mm["repeats"],
                          "frames_that_reused_the_previous_expansion": ctx.flow_cache_hits() - h0, "frames": 500}
        ctx.lk_reset()
    # cv-decoder's "Process Fullres" = false (cv-decoder/src/lib.rs:124-133,274-276): the 1080p frame -- BGR as VideoCapture hands it, or luma --
    # is resized to 150 x 84 on the device, mask + Farneback run on the reduced frames, ~12.6 k records come back.  The cheap way to the
    # fixed 150 x 84 vectors of the reference's published runs (docs/report.tex:925); its own number for the OpenCV decoder: 45.679 ms per
    # frame (docs/demo.md:85, hardware and mode not stated on the slide).
    rng = np.random.default_rng(11)
    tint = rng.integers(-40, 41, (1, H, W, 3))
    bgr4 = np.clip(fr4[..., None].astype(int) + tint, 0, 255).astype(np.uint8)
    pins_bgr = [ctx.pinned_frame(H, 3 * W).reshape(H, W, 3) for _ in range(4)]
    for k in range(4):
        np.copyto(pins_bgr[k], bgr4[k])
    reduced = {}
    for name, frames, fmt in (("bgr", pins_bgr, ctx.FMT_BGR), ("luma", pins, ctx.FMT_LUMA)):
        kw = dict(levels=5, radius=6, iters=3, farneback=True, use_previous=True, reduced=True, fmt=fmt, contrast_mask=True)

        def stream_reduced(nfr):
            prev = None
            for k in range(nfr):
                t = ctx.lk_push_frame_async(frames[k % 4], **kw)
                if prev is not None:
                    ctx.lk_frame_wait(prev, outs[k & 1])
                prev = t
            return ctx.lk_frame_wait(prev, outs[nfr & 1])
        ctx.lk_reset()
        stream_reduced(8)
        with QuietGC():
            runs = []
            for _ in range(5):
                t0 = time.perf_counter()
                stream_reduced(100)
                runs.append((time.perf_counter() - t0) / 100 * 1e3)
        mm = median_min_max(runs)
        ctx.lk_reset()
        last = stream_reduced(4)                      # a fresh stream: pairs (0,1), (1,2), (2,3) with the flow carried over
        reduced[name] = {"ms_per_frame": mm["median"], "ms_per_frame_min": mm["min"], "ms_per_frame_max": mm["max"], "repeats": mm["repeats"],
                         "records_per_frame": int(len(last[0])), "grid": list(last[1]), "frames": 500}
        kept["reduced_" + name] = np.array(last[0])
        ctx.lk_reset()
    reduced["what"] = ("hip_flow with 'Process Fullres' = false: ofps_hip_lk_push_frame_async + _frame_wait, 1080p frames (BGR 6.2 MB / luma 2.1 MB) from "
                       "page-locked memory, two tickets in flight; resize INTER_LINEAR -> gray -> contrast mask + Farneback on 150 x 84 -> one record per "
                       "unmasked reduced-frame pixel; medians of 5 x 100 frames")
    reduced["reference_published_ms"] = {"value": 45.679, "source": "docs/demo.md:85 ('OpenCV'; hardware, resolution and mode not stated)"}
    kept["reduced_frames"] = (bgr4, fr4)
    ctx.close()
    lk_ms, alm_ms = per["pm3"]["lk_ms"], per["pm3"]["almeida_ms"]
    out = {"what": "BASELINE configs[2]: 1080p pair -> 3-level LK (r=4, 3 steps) -> 2,073,600 per-pixel records -> densify 150x84 "
                   "-> Almeida LSQ, device resident; two contents",
           "content": {"pm3": "+-3 px region motion (seed 11)", "pm16": "+-16 px region motion (seed 11): the SAD bench line's kind of content"},
           "per_content": per,
           # the round-3 keys, on the +-3 content (continuity)
           "lk_ms": per["pm3"]["lk_ms"], "densify_ms": per["pm3"]["densify_ms"], "almeida_ms": per["pm3"]["almeida_ms"],
           "chain_ms": per["pm3"]["chain_ms"], "Mvectors_per_s_chain": per["pm3"]["Mvectors_per_s_chain"], "reps": reps,
           "lk_tiles_computed_by_a_waiting_child_in_timed_region": lk_waits,
           "roofline_lk": {"bound": "valu_f32", "unit": "TFLOP/s", "spec_flops_per_pair": taps * LK_SPEC_FLOPS_PER_TAP,
                           "achieved": round(taps * LK_SPEC_FLOPS_PER_TAP / (lk_ms * 1e-3) / 1e12, 3), "peak": round(VALU_F32_PEAK_FLOPS / 1e12, 2),
                           "frac": round(taps * LK_SPEC_FLOPS_PER_TAP / (lk_ms * 1e-3) / VALU_F32_PEAK_FLOPS, 4),
                           "frac_pm16": per["pm16"]["lk_frac_of_f32_peak"],
                           "instr_view": {"spec_instr_lanes_per_pair": taps * LK_SPEC_INSTR_PER_TAP, "peak_T_instr_lanes_per_s": round(VALU_F32_PEAK_INSTR / 1e12, 2),
                                          "frac": round(taps * LK_SPEC_INSTR_PER_TAP / (lk_ms * 1e-3) / VALU_F32_PEAK_INSTR, 4)},
                           # the pipe that binds the rows: LDS-array cycles of the reads the spec's access pattern needs (per wave and tap one
                           # 16-byte record read = 4 cycles, per window row ten texel reads = 2 cycles each; MI355X_MICROARCH.md, LDS table),
                           # conflict-free, against one LDS array per CU at the guide's clock
                           "lds_view": {"spec_lds_cycles_per_pair": round(taps / 64 * (4 + 20 / 9)), "peak_cycles_per_s": 256 * 2.4e9,
                                        "frac": round(taps / 64 * (4 + 20 / 9) / (lk_ms * 1e-3) / (256 * 2.4e9), 4),
                                        "frac_pm16": round(taps / 64 * (4 + 20 / 9) / (per["pm16"]["lk_ms"] * 1e-3) / (256 * 2.4e9), 4)},
                           "note": "flops of the spec's window taps over all level steps (11 per tap with a fused multiply-add = 2; the structure "
                                   "tensor, pyramid and staging are not counted as useful work) / the time of the WHOLE lk_flow call, against the "
                                   "guide's f32 vector peak (157.3 TFLOP/s, fma = 2: the same unit); instr_view: the 7 instructions per tap "
                                   "against 78.6 T lane-instructions/s; lds_view: the LDS-array cycles of the spec's reads against 256 arrays at 2.4 GHz -- "
                                   "the pipe that binds the rows (measured with conflicts: ~64 % busy over the launch, ~76 % inside the rows; DESIGN.md N2)"},
           # hip_flow (farneback.hip): all streaming.  Algorithmic bytes per pair, per layer pixel: the expansion planes
           # written once (2 x 20 B); the layer's first matrices: R0 + R1 read, M written (60 B); updates 1 and 2: M read, R0 + R1 read,
           # next M written (80 B each); the last update: M read, flow written (28 B) = 288 B; + 16 B of records per frame pixel, both
           # frames read once
           "farneback_ms": per["pm3"]["farneback_ms"],
           "decoders_read_ahead": dict(decoders, what="ofps_hip_lk_push_frame_async + ofps_hip_lk_frame_wait, 1080p frames from page-locked memory, two "
                                                      "tickets in flight, contrast mask + 150 x 84 records to the host; medians of 5 x 100 frames"),
           "decoders_reduced": reduced,
           "roofline_farneback": (lambda fpx, fb: {"bound": "hbm", "unit": "GB/s", "algorithmic_bytes": fb,
                                                   "achieved": round(fb / (per["pm3"]["farneback_ms"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                   "frac": round(fb / (per["pm3"]["farneback_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                   "frac_pm16": round(fb / (per["pm16"]["farneback_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                   "note": "six layers (1080p .. 60 x 34), three updates each; the pyramid's intermediate planes are not counted"})(
               sum(((W + (1 << k) // 2) >> k) * ((H + (1 << k) // 2) >> k) for k in range(6)),
               sum(((W + (1 << k) // 2) >> k) * ((H + (1 << k) // 2) >> k) for k in range(6)) * (40 + 60 + 2 * 80 + 28) + 16 * n + 2 * n),
           "roofline_almeida": {"bound": "hbm", "unit": "GB/s", "algorithmic_bytes": 16 * n,
                                "achieved": round(16 * n / (alm_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                "frac": round(16 * n / (alm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "note": "records read once; the solve is a 30-link dependent chain, not a stream"}}
    # ---- parity + the CPU path beside it, outside the timed region
    try:
        import oracle
        thr = min(16, oracle.num_threads())
        cam = oracle.camera(W / H, 39.6 * H / W)
        pcs, cpu = {}, {}
        for name, fr in contents.items():
            ent, q, fld = kept[name]
            if name == "pm3":                                      # N2 on the host: one thread (how the reference runs a decoder) and all the cores
                prev_thr = oracle.set_num_threads(1)
                t0 = time.perf_counter(); oracle.lk_flow(fr[0], fr[1], LV, RAD, IT); cpu["lk_flow_1_thread_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                oracle.set_num_threads(prev_thr)
            t0 = time.perf_counter()
            flow_o = oracle.lk_flow(fr[0], fr[1], LV, RAD, IT)
            cpu[f"lk_flow_all_cores_ms_{name}"] = round((time.perf_counter() - t0) * 1e3, 1)
            ent_o = oracle.flow_to_entries(flow_o)
            q_o = oracle.solve_ypr_given(ent_o, cam, threads=thr)
            if name == "pm3":                                      # hip_flow's records against the Farneback restatement (all cores)
                t0 = time.perf_counter()
                fb_o = oracle.flow_to_entries(oracle.farneback_flow(fr[0], fr[1]))
                cpu["farneback_all_cores_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                d = np.abs(kept["farneback_pm3"] - fb_o)
                farneback_pc = {"records_bit_exact": bool((kept["farneback_pm3"].view(np.uint32) == fb_o.view(np.uint32)).all()),
                                "max_abs_diff_normalised": float(d.max()), "tolerance_px": 1e-4,
                                "ok": bool((d[:, 2] * W).max() <= 1e-4 and (d[:, 3] * H).max() <= 1e-4)}
            pc = {"lk_records_bit_exact": bool((ent.view(np.uint32) == ent_o.view(np.uint32)).all()),
                  "densify_field_bit_exact": bool((fld.view(np.uint32).reshape(-1) == oracle.densify(ent_o, GW, GH).view(np.uint32).reshape(-1)).all()),
                  "almeida_max_abs_dq": float(np.abs(q - q_o).max()), "almeida_tolerance": 2e-6}
            pc["ok"] = bool(pc["lk_records_bit_exact"] and pc["densify_field_bit_exact"] and pc["almeida_max_abs_dq"] <= 2e-6)
            pcs[name] = pc
        # the reduced decoders' last frame against the oracle chain (front-end -> Farneback with the carried flow -> mask -> records)
        reduced_pc = {}
        for name, frames_o, fmt_o in (("bgr", kept["reduced_frames"][0], oracle.FMT_BGR), ("luma", kept["reduced_frames"][1], oracle.FMT_LUMA)):
            flow_r = None
            t0 = time.perf_counter()
            for k in range(1, 4):
                rec_o, _, flow_r = oracle.cv_decode(frames_o[k - 1], frames_o[k], fmt_o, process_fullres=False, init=flow_r)
            cpu[f"reduced_decoder_{name}_all_cores_ms_per_frame"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
            got = kept["reduced_" + name]
            reduced_pc[name] = bool(got.shape == rec_o.shape and (got.view(np.uint32) == rec_o.view(np.uint32)).all())
        cpu.update({"threads_all_cores": oracle.num_threads(), "kind": "port (oracle/ofps_oracle.c:orc_lk_flow, OpenMP over rows)",
                    "speedup_vs_all_cores_pm3": round(cpu["lk_flow_all_cores_ms_pm3"] / per["pm3"]["lk_ms"], 1)})
        out["cpu_lk"] = cpu
        out["parity_check"] = dict(pcs["pm3"], per_content=pcs, farneback=farneback_pc, reduced_decoders_bit_exact=reduced_pc,
                                   ok=bool(all(v["ok"] for v in pcs.values()) and farneback_pc["ok"] and all(reduced_pc.values())))
    except ImportError as e:
        out["parity_check"] = {"ok": None, "skipped": str(e)}
    return out


def cfg4_leg(device: int = 0, steps: int = 8, warmup: int = 2) -> dict:
    import torch
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    W, H, B, R, P, G = 3840, 2160, 8, 32, 64, 4
    nblk = (W // B) * (H // B)
    gen = synth.luma_sequence(G + 1, W, H, max_step=R, seed=synth.SEED0 + 4)
    walk = np.abs(((np.arange(P + 1) + G) % (2 * G)) - G)                 # every consecutive pair is a generated pair
    ctx = HipContext(device)
    ctx.use_torch_stream()
    d_frames = torch.from_numpy(gen[walk]).cuda().contiguous()
    d_out = torch.empty((P, nblk, 4), dtype=torch.float32, device="cuda")
    d_best = torch.empty((P, nblk, 3), dtype=torch.int32, device="cuda")

    def step(best=False):
        ctx.sad_flow_dev(d_frames.data_ptr(), P + 1, W, H, W, W * H, 0, B, R, d_out.data_ptr(), d_best.data_ptr() if best else None)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    launch_ms = _event_ms(ctx, step, steps, warm=0)
    absd = P * nblk * B * B * (2 * R + 1) ** 2
    algo = P * (2 * W * H + 16 * nblk)
    out = {"what": "BASELINE configs[3] on one GPU: 3840x2160, 8x8 blocks, +-32 full search, one resident batch of 64 pairs",
           "Mvectors_per_s": round(P * nblk * steps / el / 1e6, 2), "ms_per_step": round(el / steps * 1e3, 3),
           "ms_per_frame_pair": round(el / steps / P * 1e3, 4), "steps": steps, "pairs_per_step": P, "vectors_per_pair": nblk,
           "kernel": "sad_strip_kernel<8,32>", "launch_ms": round(launch_ms, 3),
           "valu": {"abs_diffs_per_launch": absd, "achieved_Tops": round(absd / (launch_ms * 1e-3) / 1e12, 2),
                    "peak_Tops": round(SAD_PEAK / 1e12, 2), "frac": round(absd / (launch_ms * 1e-3) / SAD_PEAK, 4)},
           "hbm": {"algorithmic_bytes_per_launch": algo, "achieved_GBs": round(algo / (launch_ms * 1e-3) / 1e9, 1),
                   "frac": round(algo / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
    # ---- parity: an exhaustive search is local, so a block whose +-R window lies inside an aligned crop gets the same
    # (dx, dy, SAD) from the oracle run on the crop alone (the full 4K pair would take the oracle half a minute)
    step(best=True)
    torch.cuda.synchronize()
    try:
        import oracle
        ok = True
        for k, (cx, cy, cw, ch) in ((0, (0, 0, 256, 160)), (P - 1, (W - 256, H - 160, 256, 160)), (P // 2, (1792, 1000, 256, 160))):
            best = d_best[k].cpu().numpy().reshape(H // B, W // B, 3)
            a, b = gen[walk[k]], gen[walk[k + 1]]
            _, bo = oracle.sad_flow(a[cy:cy + ch, cx:cx + cw], b[cy:cy + ch, cx:cx + cw], B, R, threads=8)
            bo = bo.reshape(ch // B, cw // B, 3)
            x_lo = 0 if cx == 0 else R // B; x_hi = cw // B if cx + cw == W else cw // B - R // B
            y_lo = 0 if cy == 0 else R // B; y_hi = ch // B if cy + ch == H else ch // B - R // B
            got = best[cy // B + y_lo:cy // B + y_hi, cx // B + x_lo:cx // B + x_hi]
            ok = ok and bool((got == bo[y_lo:y_hi, x_lo:x_hi]).all())
        out["parity_check"] = {"what": "three pairs of the batch, blocks of an aligned crop vs the oracle on the crop, (dx, dy, SAD) exact",
                               "ok": ok}
    except ImportError as e:
        out["parity_check"] = {"ok": None, "skipped": str(e)}
    ctx.close()
    return out


def _recv_exact(sock, n, buf):
    view = memoryview(buf)[:n]
    got = 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise EOFError
        got += r


def cfg5_stream_leg(device: int = 0, frames: int = 330, fps: float = 60.0, use_ransac: bool = False) -> dict:
    """The feeder thread paces raw luma frames (8-byte header: u32 W, u32 H, then W*H bytes) through a loop-back socket, the
    way ofps::utils::open_file("tcp://...") feeds a decoder (ofps/src/utils.rs:92-118); latency = frame fully received ->
    island + quaternion on the host."""
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    W, H = 1920, 1080
    clip = synth.luma_sequence(8, W, H, max_step=16)              # looped
    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    port = srv.getsockname()[1]

    def feeder():
        c, _ = srv.accept()
        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        t_next = time.perf_counter()
        for k in range(frames):
            now = time.perf_counter()
            if now < t_next:
                time.sleep(t_next - now)
            t_next += 1.0 / fps
            c.sendall(struct.pack("<II", W, H)); c.sendall(clip[k % len(clip)].tobytes())
        c.close()
    th = threading.Thread(target=feeder, daemon=True); th.start()
    sock = socket.create_connection(("127.0.0.1", port))
    ctx = HipContext(device)
    pinned = ctx.pinned_frame(H, W)                               # recv_into writes the frame where the DMA engine reads it
    buf = memoryview(pinned).cast("B")
    lat, islands, results = [], 0, {}
    check_at = (frames - 3, frames - 2)                           # frames whose results are compared with the oracle afterwards
    hdr = bytearray(8)
    for k in range(frames):
        _recv_exact(sock, 8, hdr)
        w, h = struct.unpack("<II", hdr)
        _recv_exact(sock, w * h, buf)
        t_arr = time.perf_counter()
        r = ctx.push_frame(pinned, block=16, search_range=16, aspect=W / H, fov_y_deg=39.6 * H / W, use_ransac=use_ransac, seed=k)
        t_done = time.perf_counter()
        if k >= 10:
            lat.append((t_done - t_arr) * 1e3)
        islands += r["motion"] is not None
        if k in check_at:
            results[k] = r
    lat = np.array(lat)
    recov = ctx.almeida_recoveries()
    ctx.close()
    out = {"what": f"BASELINE configs[4]: {W}x{H}@{fps:g} over loop-back TCP, 16x16 +-16 SAD + block-motion + almeida "
                   f"({'RANSAC' if use_ransac else 'LSQ'}) fused per frame; frame received -> island + quaternion on the host",
           "frames": frames, "measured_frames": int(len(lat)),
           "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "p90": round(float(np.percentile(lat, 90)), 3),
                          "p99": round(float(np.percentile(lat, 99)), 3), "max": round(float(lat.max()), 3)},
           "frame_budget_ms": round(1e3 / fps, 2), "frames_with_motion_island": int(islands),
           "almeida_in_kernel_recoveries": int(recov)}
    try:
        import oracle
        cam = oracle.camera(W / H, 39.6 * H / W)
        ok = True
        worst = 0.0
        for k, r in results.items():
            ent_o, _ = oracle.sad_flow(clip[(k - 1) % len(clip)], clip[k % len(clip)], 16, 16, threads=4)
            q_o = oracle.solve_ypr_given(ent_o, cam) if not use_ransac else oracle.solve_ypr_ransac(ent_o, cam, 200, 0.05, 1000, seed=k)
            det_o = oracle.detect_motion(ent_o)
            worst = max(worst, float(np.abs(r["quat"] - q_o).max()))
            ok = ok and (r["motion"] is None) == (det_o is None) and (det_o is None or r["motion"][0] == det_o[0])
        tol = 1e-4 if use_ransac else 2e-6
        out["parity_check"] = {"what": "two streamed frames: island area exact, quaternion vs the oracle on the oracle's vectors",
                               "max_abs_dq": worst, "tolerance": tol, "ok": bool(ok and worst <= tol)}
    except ImportError as e:
        out["parity_check"] = {"ok": None, "skipped": str(e)}
    return out


def _cfg5_processes(use_ransac: bool, n: int = 3, frames: int = 160):
    """p50 of n FRESH processes (host + loop-back TCP + PCIe latency depends on where the process landed: cores, page placement, the
    runtime's DMA engine binding -- profiles/r05/batched_bimodal.txt -- and on what the host's other tenants are doing); the perf gate compares
    the best of them (round 6)"""
    import subprocess
    p50s = []
    for _ in range(n):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "cfg5_once_ransac" if use_ransac else "cfg5_once_lsq", str(frames)],
                           capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            return {"error": (p.stderr or p.stdout)[-300:]}
        r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        p50s.append(r["latency_ms"]["p50"])
    mm = median_min_max(p50s, 3)
    return {"p50_median_of_processes": mm["median"], "p50_min": mm["min"], "p50_max": mm["max"], "processes": n, "frames_per_process": frames}


def cfg5_both_leg(device: int = 0) -> dict:
    """cfg5 with the estimator both ways: LSQ (the top-level keys, as in round 3) and the reference's DEFAULT, RANSAC with 200
    hypotheses x 1000 samples (almeida-estimator/src/lib.rs:69-78: use_ransac = true), per-frame seeds, parity per seed."""
    out = cfg5_stream_leg(device, use_ransac=False)
    r = cfg5_stream_leg(device, frames=210, use_ransac=True)
    out["process_level"] = {"lsq": _cfg5_processes(False), "ransac": _cfg5_processes(True),
                            "what": "p50 of three fresh processes each (160 frames at 60 Hz); tools/perf_gate.py gates the best of the three (p50_min)"}
    out["ransac"] = {k: r[k] for k in ("what", "frames", "measured_frames", "latency_ms", "parity_check", "almeida_in_kernel_recoveries")}
    out["lsq"] = {"latency_ms": out["latency_ms"], "parity_check": out["parity_check"]}
    out["parity_check"] = dict(out["parity_check"], ok=(None if out["parity_check"].get("ok") is None or r["parity_check"].get("ok") is None
                                                         else bool(out["parity_check"]["ok"] and r["parity_check"]["ok"])),
                               ransac_max_abs_dq=r["parity_check"].get("max_abs_dq"))
    return out


def all_legs(device: int = 0) -> dict:
    out = {}
    for name, fn in (("cfg3_chain", cfg3_chain_leg), ("cfg4", cfg4_leg), ("cfg5_stream", cfg5_both_leg)):
        t0 = time.perf_counter()
        try:
            with QuietGC() as quiet:                            # host-timed loops inside: see QuietGC
                out[name] = fn(device)
            out[name]["python_gc_inside_leg"] = quiet.summary()
        except Exception as e:                                  # a leg is evidence beside the bench line, never the line itself
            out[name] = {"error": repr(e)[:300]}
        out[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return out


if __name__ == "__main__":
    only = sys.argv[1:] or None
    legs = {"cfg3_chain": cfg3_chain_leg, "cfg4": cfg4_leg, "cfg5_stream": cfg5_both_leg}
    if only and only[0] in ("cfg5_once_lsq", "cfg5_once_ransac"):          # one process of cfg5_both_leg's process-level repeats
        with QuietGC():
            r = cfg5_stream_leg(0, frames=int(only[1]) if len(only) > 1 else 160, use_ransac=only[0].endswith("ransac"))
        print(json.dumps({"latency_ms": r["latency_ms"], "parity_ok": r["parity_check"].get("ok")}), flush=True)
        sys.exit(0)
    if only:
        res = {}
        for nme in only:
            t0 = time.perf_counter()
            with QuietGC() as quiet:
                res[nme] = legs[nme](0)
            res[nme]["python_gc_inside_leg"] = quiet.summary()
            res[nme]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    else:
        res = all_legs(0)
    print(json.dumps(res), flush=True)
