#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory from the CPU oracle (oracle/ofps_oracle.c).

The reference is Rust and cannot be run here (no rustc/cargo), so these are NOT outputs
of the reference: they are the oracle's outputs on fixed seeded inputs, committed so that (a) the oracle
cannot drift silently, (b) the HIP path is checked against fixed bytes as well as against a live oracle.
The Almeida inputs reproduce the reference's own test construction (almeida-estimator/src/lib.rs:253-306);
its acceptance bound (error < 10 % of the rotation) is asserted when the file is generated.

Run from the repo root:  python tests/golden/generate.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from ofps_amd import synth  # noqa: E402
import almeida_cases as ac  # noqa: E402


def entries(n, seed, lo=0.0, hi=1.0, scale=0.01):
    rng = np.random.default_rng(seed)
    e = np.empty((n, 4), np.float32)
    e[:, :2] = rng.uniform(lo, hi, (n, 2)).astype(np.float32)
    e[:, 2:] = rng.normal(0, scale, (n, 2)).astype(np.float32)
    return e


def frontend():
    """frontend.npz (round 6): cv-decoder's frame front-end and its "Process Fullres" = false mode on one small BGR pair -- the resized colour
    frame, the gray frames both modes leave, the reduced decoder's records with Farneback's flow (cv-decoder's own) cold and with the
    previous flow carried over, and with the build's LK flow.  python tests/golden/generate.py frontend"""
    rng = np.random.default_rng(61)
    y = synth.flatten_regions(synth.luma_sequence(3, 320, 180, max_step=3, seed=61), region=36, seed=62)
    # (channel offsets + a little noise: the flattened regions stay flat enough for the contrast mask to drop them)
    bgr = np.clip(y[..., None].astype(int) + np.array([12, -7, 20]) + rng.integers(-2, 3, (1, 180, 320, 3)), 0, 255).astype(np.uint8)
    gw, gh = oracle.cv_grid(320, 180, 120, 120)
    d = {"bgr": bgr, "grid_120": np.array([gw, gh]), "bgr_small": oracle.resize_linear(bgr[1], gw, gh),
         "gray_full": oracle.cv_frontend(bgr[1], oracle.FMT_BGR), "gray_small": oracle.cv_frontend(bgr[1], oracle.FMT_BGR, False, 120, 120),
         "luma_small": oracle.resize_linear(y[1], gw, gh)}
    r1, _, f1 = oracle.cv_decode(bgr[0], bgr[1], oracle.FMT_BGR, process_fullres=False, max_w=120, max_h=120)
    r2, _, _ = oracle.cv_decode(bgr[1], bgr[2], oracle.FMT_BGR, process_fullres=False, max_w=120, max_h=120, init=f1)
    r_lk, _, _ = oracle.cv_decode(bgr[0], bgr[1], oracle.FMT_BGR, process_fullres=False, max_w=120, max_h=120, flow="lk")
    assert 0.2 * gw * gh < len(r1) < 0.98 * gw * gh, len(r1)              # partly masked
    d.update(records_pair01=r1, flow_pair01=f1, records_pair12_warm=r2, records_pair01_lk=r_lk)
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **d)
    print({k: v.shape for k, v in d.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "frontend":
        return frontend()
    # ---- Almeida: 8 of the reference's 32 cases with full fields (inputs are 2.5k x 4 floats each), all 32 answers
    cam = oracle.camera(1.0, 90.0)
    cases = ac.cases()
    q_true = np.array([c[2] for c in cases], np.float32)
    q_lsq = np.array([oracle.solve_ypr_given(c[3], cam) for c in cases], np.float32)
    q_ransac = np.array([oracle.solve_ypr_ransac(c[3], cam, 100, 0.05, 1000, seed=1234 + i) for i, c in enumerate(cases)], np.float32)
    for (rot, ang, q, f), ql, qr in zip(cases, q_lsq, q_ransac):
        assert ac.error_deg(q, ql) < 0.1 * rot or ac.error_deg(q, ql) == 0
        assert ac.error_deg(q, qr) < 0.1 * rot or ac.error_deg(q, qr) == 0
    keep = [1, 7, 12, 15, 19, 23, 28, 31]
    np.savez_compressed(os.path.join(HERE, "almeida.npz"), rot=np.array([c[0] for c in cases], np.float32),
                        q_true=q_true, q_lsq=q_lsq, q_ransac=q_ransac, ransac_seed0=np.int64(1234),
                        field_index=np.array(keep), **{f"field_{i}": cases[i][3] for i in keep})

    # ---- densifier: seeded entries on the detector's default grid and cv-decoder's 150x84; an edge-case set
    d = {}
    for name, (n, w, h, seed) in {"a": (1000, 14, 14, 1), "b": (1000, 150, 84, 2), "c": (257, 7, 3, 3)}.items():
        e = entries(n, seed)
        f, cells = oracle.densify(e, w, h, want_cells=True)
        d[f"in_{name}"], d[f"wh_{name}"], d[f"field_{name}"], d[f"cells_{name}"] = e, np.array([w, h]), f, cells
    e = entries(48, 4, -0.5, 1.5)
    e[0, :2] = (np.nan, 0.5); e[1, :2] = (0.0, 0.5); e[2, :2] = (1.0, 0.2); e[3, :2] = (np.inf, 0.1)
    f, cells = oracle.densify(e, 9, 5, want_cells=True)      # clamp semantics: SURVEY A.6, unverified vs rustc
    d["in_edge"], d["wh_edge"], d["field_edge"], d["cells_edge"] = e, np.array([9, 5]), f, cells
    np.savez_compressed(os.path.join(HERE, "densify.npz"), **d)

    # ---- detector: random fields (Some and None outcomes) + a hand-built tie/seed case
    g = {}
    for k, (seed, scale) in enumerate([(10, 0.3), (11, 30.0), (12, 0.7), (13, 1.0)]):
        e = entries(4000, seed); e[:, 2:] *= np.float32(scale)
        r = oracle.detect_motion(e)
        g[f"in_{k}"] = e
        g[f"some_{k}"] = np.int32(r is not None)
        g[f"area_{k}"] = np.int64(r[0] if r else 0)
        g[f"field_{k}"] = r[1] if r else np.zeros((14, 14, 2), np.float32)
    np.savez_compressed(os.path.join(HERE, "detect.npz"), **g)

    # ---- SAD: small synthetic pairs with planted displacements, full (dx,dy,SAD) tables
    s = {}
    for name, (W, H, B, R) in {"64x48_b16_r8": (64, 48, 16, 8), "640x360_b16_r8": (640, 360, 16, 8),
                               "256x144_b16_r16": (256, 144, 16, 16), "160x96_b8_r32": (160, 96, 8, 32)}.items():
        fr = synth.luma_sequence(2, W, H, max_step=R, seed=synth.SEED0 + W)
        ent, best = oracle.sad_flow(fr[0], fr[1], B, R, simd=False)
        s[f"frames_{name}"], s[f"best_{name}"], s[f"entries_{name}"] = fr, best, ent
    np.savez_compressed(os.path.join(HERE, "sad.npz"), **s)
    # ---- dense flow decoder (N2 + cv-decoder's mask / output stage): 160x96 pair with flat regions
    fr = synth.flatten_regions(synth.luma_sequence(2, 160, 96, max_step=2, seed=synth.SEED0 + 5), region=32, seed=4)
    flow = oracle.lk_flow(fr[0], fr[1], 3, 4, 3)
    mask = oracle.contrast_mask(fr[1])
    rec = oracle.masked_flow_to_entries(flow, mask)
    np.savez_compressed(os.path.join(HERE, "flow.npz"), frames=fr, flow=flow, mask=mask, records=rec,
                        cells_60x36=oracle.densify_to_entries(rec, 60, 36))
    flow_revisions(fr)
    # ---- cfg3-sized Almeida LSQ: 1920x1080 per-pixel records (inputs are regenerated from synth; 18 s of oracle time)
    e = synth.rotation_field(1920, 1080)
    np.savez_compressed(os.path.join(HERE, "almeida_dense.npz"), q_lsq=oracle.solve_ypr_given(e, oracle.camera(16 / 9, 22.275)),
                        checksum=np.float64(e.astype(np.float64).sum()), first=e[:4], last=e[-4:])
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


def flow_revisions(fr=None):
    """flow_rev1.npz: the N2 spec's two revisions on one small pair (a 64 x 40 crop of flow.npz's frames) -- revision 1 (separate
    multiply and add) from the numpy restatement with its switch off, revision 2 (fused, the shipped spec) from the C oracle.
    The test that reads it fails if the shipped spec drifts back to revision 1 arithmetic, or away from both."""
    from oracle import np_oracle
    if fr is None:
        fr = synth.flatten_regions(synth.luma_sequence(2, 160, 96, max_step=2, seed=synth.SEED0 + 5), region=32, seed=4)
    crop = np.ascontiguousarray(fr[:, 20:60, 40:104])
    np_oracle.LK_SPEC_FMA = False
    try:
        rev1 = np_oracle.lk_flow(crop[0], crop[1], 2, 4, 3)
    finally:
        np_oracle.LK_SPEC_FMA = True
    rev2 = oracle.lk_flow(crop[0], crop[1], 2, 4, 3)
    assert oracle.lk_spec_revision() == 2 and (rev1.view(np.uint32) != rev2.view(np.uint32)).any()
    np.savez_compressed(os.path.join(HERE, "flow_rev1.npz"), frames=crop, flow_rev1=rev1, flow_rev2=rev2)


if __name__ == "__main__":
    main()
