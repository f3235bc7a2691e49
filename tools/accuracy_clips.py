#!/usr/bin/env python3
"""End-to-end accuracy in the reference's own unit: mean / max rotation error per frame, in degrees, of the
image -> vectors -> rotation path against PLANTED camera rotations -- what docs/statistics/err_av.csv tabulates for the reference's
estimators against Blender ground truth (loader ofps-suite/src/app/tracking/mod.rs:125-217, export :825-847, summary
scripts/extract_stats.py: degs(error.mean())).

Clips are rendered by ofps_amd.synth.rotation_clip from per-frame rotations through the pinhole model of StandardCamera (pan / tilt /
roll / mixed, 0.01 - 1 degree per frame, one clip with an independently moving foreground object = the reference's "dyn" clips).
Every clip runs through the tracking loop of ofps-suite/src/app/tracking/worker.rs:305-412 written with the plugin mirrors
(ofps_amd/plugins.py): decoder.process_frame -> estimator.motion_step (pose accumulation, ofps/src/estimator.rs:38-53), for
  decoders    hip_sad (16x16 blocks, +-16)  |  hip_lk (3-level pyramid, r = 4, 3 steps, contrast mask, 150 x 84 records)  |  hip_lk5 (the same
              with "Pyramid levels" = 5, the reference's Farneback depth)  |  hip_flow (Farneback's polynomial-expansion flow with
              cv-decoder's own arguments: levels 5, winsize 13, 3 iterations, poly_n 7, poly_sigma 1.5; same mask and records)
  estimators  hip_almeida LSQ  |  hip_almeida RANSAC (the reference's default: 200 hypotheses x 1000 samples, 0.05 degree inliers)
Per clip and combination: mean and max of angle_to(planted q_k, estimated r_k) over the frames, that mean relative to the clip's mean
rotation per frame (the reference's own test bound is 10 %: almeida-estimator/src/lib.rs:347-348), and the pose drift after the
last frame.  Beside it the CPU oracle chain on the same frames (all frames for hip_sad + LSQ; the first `--oracle-lk-pairs` pairs
for hip_lk + LSQ) and the largest |dq| between the two chains -- the oracle is the checker here, never the thing measured.

  python tools/accuracy_clips.py [--quick] [--out profiles/r05/accuracy.txt] [--json gpurun_out/r05/accuracy.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ofps_amd import synth  # noqa: E402


def triangle(rate, n, period):
    """+rate for period frames, -rate for 2 x period, +rate ... : constant speed, bounded accumulated angle."""
    k = np.arange(n)
    return np.where(((k + period) // (2 * period)) % 2 == 0, rate, -rate).astype(np.float64)


def clip_table(quick=False):
    """name -> (W, H, fov_y_deg, per-frame eulers [n, 3] (roll = tilt, pitch = image roll, yaw = pan), distractor)"""
    n = 24 if quick else 60
    W, H = 1920, 1080
    z = np.zeros(n)
    k = np.arange(n)
    t = {}
    t["pan_0.2"] = (W, H, 60.0, np.stack([z, z, triangle(0.2, n, 20)], 1), None)
    t["tilt_0.2"] = (W, H, 60.0, np.stack([triangle(0.2, n, 20), z, z], 1), None)
    t["roll_0.3"] = (W, H, 60.0, np.stack([z, triangle(0.3, n, 20), z], 1), None)
    t["slow_pan_0.01"] = (W, H, 60.0, np.stack([z, z, z + 0.01], 1), None)
    t["pan_tilt_0.05"] = (W, H, 60.0, np.stack([z + 0.05, z, z - 0.05], 1), None)
    t["fast_pan_0.5"] = (W, H, 60.0, np.stack([z, z, triangle(0.5, n, 12)], 1), None)
    mixed = np.stack([0.25 * np.sin(2 * np.pi * k / 30.0), 0.3 * np.sin(2 * np.pi * k / 24.0 + 1.0), 0.3 * np.cos(2 * np.pi * k / 40.0)], 1)
    t["mixed_sine"] = (W, H, 60.0, mixed, None)
    t["mixed_sine_dyn"] = (W, H, 60.0, mixed, dict(size=(W // 5, H // 4), start=(W // 10, H // 3), velocity=(W / 1920 * 7.0, -H / 1080 * 2.0)))
    if not quick:
        t["1deg_mixed_360p"] = (640, 360, 60.0, np.stack([triangle(0.6, n, 8), triangle(0.3, n, 10), triangle(0.7, n, 6)], 1), None)
    return t


def track(frames, cam, decoder_cls, use_ransac, device=0, **dec_props):
    """The tracking worker's loop (worker.rs:305-412): -> (per-pair quaternions [n, 4], accumulated rotation, seconds per frame)."""
    from ofps_amd.plugins import HipAlmeidaEstimator
    dec = decoder_cls(iter(frames), device=device)
    for name, v in dec_props.items():
        assert dec.set_prop(name, v), name
    est = HipAlmeidaEstimator(device)
    est.set_prop("Use ransac", use_ransac)
    rot, pos = np.array([1.0, 0, 0, 0], np.float32), np.zeros(3, np.float32)
    out = []
    t0 = time.perf_counter()
    while True:
        field = []
        try:
            have = dec.process_frame(field)
        except EOFError:
            break
        if not have:
            continue
        new_rot, pos = est.motion_step(field, cam, None, rot, pos)     # estimator.rs:38-53: rot = r * rot, pos += rot * tr (tr = 0)
        r = _qmul64(new_rot, [rot[0], -rot[1], -rot[2], -rot[3]])      # r = new_rot * rot^-1, the estimate of this pair ...
        out.append(r / np.linalg.norm(r))                              # ... up to |rot|^2: f32 products of f32 unit quaternions drift in norm (~3e-7 per frame), as nalgebra's do
        rot = new_rot
    dt = (time.perf_counter() - t0) / max(1, len(out))
    dec.ctx.close(); est.ctx.close()
    return np.array(out), rot, dt


def _qmul(a, b):
    from ofps_amd.plugins import quat_mul
    return quat_mul(a, b)


def _qmul64(a, b):
    aw, ai, aj, ak = [float(v) for v in a]
    bw, bi, bj, bk = [float(v) for v in b]
    return np.array([aw * bw - ai * bi - aj * bj - ak * bk, aw * bi + ai * bw + aj * bk - ak * bj,
                     aw * bj - ai * bk + aj * bw + ak * bi, aw * bk + ai * bj - aj * bi + ak * bw])


def accumulate(quats):
    rot = np.array([1.0, 0, 0, 0], np.float32)
    for q in quats:
        rot = _qmul(np.asarray(q, np.float32), rot)
    return rot


def stats(est, truth):
    err = np.array([synth.quat_angle_deg(e, t) for e, t in zip(est, truth)])
    rate = np.array([synth.quat_angle_deg(t, [1, 0, 0, 0]) for t in truth])
    return {"mean_err_deg": float(err.mean()), "max_err_deg": float(err.max()), "mean_rot_deg": float(rate.mean()),
            "rel_mean": float(err.mean() / rate.mean()) if rate.mean() > 0 else None,
            "drift_deg": synth.quat_angle_deg(accumulate(est), accumulate(truth))}


def run(quick=False, oracle_lk_pairs=3, with_oracle=True, only=None, log=print):
    from ofps_amd.plugins import HipFlowDecoder, HipLkDecoder, HipSadDecoder, StandardCamera
    combos = [("hip_sad", HipSadDecoder, False, {}), ("hip_sad", HipSadDecoder, True, {}), ("hip_lk", HipLkDecoder, False, {}),
              ("hip_lk", HipLkDecoder, True, {}),
              # the reference's dense decoder runs Farneback over FIVE pyramid levels (cv-decoder/src/lib.rs:188-199): the same decoder with
              # its "Pyramid levels" property at 5 (capture range x4)
              ("hip_lk5", HipLkDecoder, False, {"Pyramid levels": 5}), ("hip_lk5", HipLkDecoder, True, {"Pyramid levels": 5}),
              # Farneback's flow itself with cv-decoder's arguments (the hip_flow decoder)
              ("hip_flow", HipFlowDecoder, False, {}), ("hip_flow", HipFlowDecoder, True, {})]
    res = {}
    for name, (W, H, fov, eul, dis) in clip_table(quick).items():
        if only and name not in only:
            continue
        t0 = time.perf_counter()
        frames, truth = synth.rotation_clip(eul, W, H, fov, seed=21 + len(res), distractor=dis)
        t_render = time.perf_counter() - t0
        cam = StandardCamera(W / H, fov)
        row = {"geometry": f"{W}x{H}", "fov_y_deg": fov, "frames": len(frames), "px_per_deg_at_centre": round(H / 2 / np.tan(np.radians(fov) / 2) * np.radians(1.0), 2),
               "dynamic_object": bool(dis), "render_s": round(t_render, 1)}
        per_pair = {}
        for dname, dcls, ransac, props in combos:
            q, rot, dt = track(frames, cam, dcls, ransac, **props)
            assert len(q) == len(truth), (len(q), len(truth))
            key = f"{dname}+{'ransac' if ransac else 'lsq'}"
            row[key] = dict(stats(q, truth), ms_per_frame=round(dt * 1e3, 3))
            per_pair[key] = q
        if with_oracle:
            import oracle
            ocam = oracle.camera(W / H, fov)
            qo = np.array([oracle.solve_ypr_given(oracle.sad_flow(frames[k], frames[k + 1], 16, 16, threads=min(16, oracle.num_threads()))[0], ocam)
                           for k in range(len(truth))])
            row["cpu_oracle:sad+lsq"] = dict(stats(qo, truth), max_abs_dq_vs_hip=float(np.abs(qo - per_pair["hip_sad+lsq"]).max()))
            m = min(oracle_lk_pairs, len(truth))
            ql = []
            for k in range(m):
                flow = oracle.lk_flow(frames[k], frames[k + 1], 3, 4, 3)
                ent = oracle.densify_to_entries(oracle.masked_flow_to_entries(flow, oracle.contrast_mask(frames[k + 1])), 150, 84)
                ql.append(oracle.solve_ypr_given(ent, ocam))
            ql = np.array(ql)
            row["cpu_oracle:lk+lsq"] = {"pairs": m, "mean_err_deg": float(np.mean([synth.quat_angle_deg(a, b) for a, b in zip(ql, truth[:m])])),
                                        "max_abs_dq_vs_hip": float(np.abs(ql - per_pair["hip_lk+lsq"][:m]).max())}
        res[name] = row
        log(f"[accuracy] {name}: " + "  ".join(f"{k} {v['mean_err_deg']:.4f}" for k, v in row.items() if isinstance(v, dict) and "mean_err_deg" in v))
    return res


def table(res):
    cols = ["hip_sad+lsq", "hip_sad+ransac", "hip_lk+lsq", "hip_lk+ransac", "hip_lk5+lsq", "hip_lk5+ransac", "hip_flow+lsq", "hip_flow+ransac",
            "cpu_oracle:sad+lsq"]
    lines = []
    lines.append("mean rotation error per frame, degrees (docs/statistics/err_av.csv's unit); clip rows, decoder+estimator columns")
    lines.append("clip,geometry,mean_rot_deg_per_frame,px_per_deg," + ",".join(cols))
    for name, r in res.items():
        lines.append(f"{name},{r['geometry']},{r['hip_sad+lsq']['mean_rot_deg']:.4f},{r['px_per_deg_at_centre']}," +
                     ",".join(f"{r[c]['mean_err_deg']:.5f}" if c in r else "" for c in cols))
    lines.append("")
    lines.append("the same as a fraction of the clip's mean rotation per frame (the reference's unit-test bound: < 0.10, almeida-estimator/src/lib.rs:347-348)")
    lines.append("clip," + ",".join(cols))
    for name, r in res.items():
        lines.append(f"{name}," + ",".join(f"{r[c]['rel_mean']:.4f}" if c in r and r[c].get("rel_mean") is not None else "" for c in cols))
    lines.append("")
    lines.append("max error of a single frame, degrees")
    lines.append("clip," + ",".join(cols))
    for name, r in res.items():
        lines.append(f"{name}," + ",".join(f"{r[c]['max_err_deg']:.5f}" if c in r else "" for c in cols))
    lines.append("")
    lines.append("pose drift after the last frame, degrees (motion_step accumulation, ofps/src/estimator.rs:38-53, vs the planted rotations accumulated the same way)")
    lines.append("clip,frames," + ",".join(cols))
    for name, r in res.items():
        lines.append(f"{name},{r['frames']}," + ",".join(f"{r[c]['drift_deg']:.4f}" if c in r else "" for c in cols))
    lines.append("")
    lines.append("HIP chain vs CPU oracle chain on the same frames: largest |dq| component (sad+lsq: every pair; lk+lsq: the first pairs); tracking-loop ms per frame (host loop incl. PCIe)")
    lines.append("clip,sad+lsq_max_abs_dq,lk+lsq_max_abs_dq,lk_pairs_checked,ms_per_frame_hip_sad+lsq,ms_per_frame_hip_lk+lsq,ms_per_frame_hip_sad+ransac")
    for name, r in res.items():
        a, b = r.get("cpu_oracle:sad+lsq", {}), r.get("cpu_oracle:lk+lsq", {})
        lines.append(f"{name},{a.get('max_abs_dq_vs_hip', float('nan')):.2e},{b.get('max_abs_dq_vs_hip', float('nan')):.2e},{b.get('pairs', 0)},"
                     f"{r['hip_sad+lsq']['ms_per_frame']},{r['hip_lk+lsq']['ms_per_frame']},{r['hip_sad+ransac']['ms_per_frame']}")
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="24 frames per clip instead of 60 (what the -m gpu test runs)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--oracle-lk-pairs", type=int, default=3)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--only", nargs="*")
    args = ap.parse_args()
    res = run(args.quick, args.oracle_lk_pairs, not args.no_oracle, args.only, log=lambda s: print(s, file=sys.stderr, flush=True))
    txt = table(res)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(__doc__.split("\n\n  python")[0] + "\n\n" + txt)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
