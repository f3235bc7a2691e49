"""Golden vectors (tests/golden/*.npz, made by tests/golden/generate.py from the oracle): the CPU half
checks that the oracle still reproduces them, the -m gpu half checks the HIP path against the same bytes."""
import os

import numpy as np
import pytest

import oracle
from ofps_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


SAD_KEYS = [("64x48_b16_r8", 16, 8), ("640x360_b16_r8", 16, 8), ("256x144_b16_r16", 16, 16), ("160x96_b8_r32", 8, 32)]


# ---------------------------------------------------------------- oracle vs golden (CPU)
def test_oracle_reproduces_almeida_golden():
    g = _load("almeida.npz")
    cam = oracle.camera(1.0, 90.0)
    for i in g["field_index"]:
        f = g[f"field_{i}"]
        np.testing.assert_array_equal(oracle.solve_ypr_given(f, cam).view(np.uint32), g["q_lsq"][i].view(np.uint32))
        q = oracle.solve_ypr_ransac(f, cam, 100, 0.05, 1000, seed=int(g["ransac_seed0"]) + int(i))
        np.testing.assert_array_equal(q.view(np.uint32), g["q_ransac"][i].view(np.uint32))
    # the reference's acceptance bound on all 32 stored answers (almeida-estimator/src/lib.rs:343-348)
    for rot, qt, ql, qr in zip(g["rot"], g["q_true"], g["q_lsq"], g["q_ransac"]):
        for q in (ql, qr):
            err = np.degrees(oracle.quat_angle_to(qt, q))
            assert err < 0.1 * rot or err == 0


def test_oracle_reproduces_densify_detect_sad_golden():
    g = _load("densify.npz")
    for k in ("a", "b", "c", "edge"):
        w, h = g[f"wh_{k}"]
        f, cells = oracle.densify(g[f"in_{k}"], int(w), int(h), want_cells=True)
        np.testing.assert_array_equal(cells, g[f"cells_{k}"])
        np.testing.assert_array_equal(f.view(np.uint32), g[f"field_{k}"].view(np.uint32))
    g = _load("detect.npz")
    for k in range(4):
        r = oracle.detect_motion(g[f"in_{k}"])
        assert (r is not None) == bool(g[f"some_{k}"])
        if r:
            assert r[0] == int(g[f"area_{k}"])
            np.testing.assert_array_equal(r[1].view(np.uint32), g[f"field_{k}"].view(np.uint32))
    g = _load("sad.npz")
    for name, B, R in SAD_KEYS:
        fr = g[f"frames_{name}"]
        ent, best = oracle.sad_flow(fr[0], fr[1], B, R)         # SIMD inner loop vs stored scalar-loop result
        np.testing.assert_array_equal(best, g[f"best_{name}"])
        np.testing.assert_array_equal(ent.view(np.uint32), g[f"entries_{name}"].view(np.uint32))


def test_oracle_reproduces_flow_golden():
    from oracle import np_oracle
    g = _load("flow.npz")
    fr = g["frames"]
    np.testing.assert_array_equal(oracle.lk_flow(fr[0], fr[1], 3, 4, 3).view(np.uint32), g["flow"].view(np.uint32))
    np.testing.assert_array_equal(oracle.contrast_mask(fr[1]), g["mask"])
    np.testing.assert_array_equal(np_oracle.contrast_mask(fr[1]), g["mask"])
    assert 0.1 < g["mask"].mean() < 0.9
    rec = oracle.masked_flow_to_entries(g["flow"], g["mask"])
    np.testing.assert_array_equal(rec.view(np.uint32), g["records"].view(np.uint32))
    np.testing.assert_array_equal(oracle.densify_to_entries(rec, 60, 36).view(np.uint32), g["cells_60x36"].view(np.uint32))


def test_flow_revision_1_golden_guards_the_spec_against_drift():
    """the N2 spec is build-defined, so the only guard against an accidental change of its arithmetic is a vector of
    the OTHER revision: revision 1 (separate multiply and add) is reproduced by the numpy restatement with its switch off, the
    shipped revision 2 by both restatements, and the two differ on this input."""
    from oracle import np_oracle
    g = _load("flow_rev1.npz")
    fr = g["frames"]
    np_oracle.LK_SPEC_FMA = False
    try:
        rev1 = np_oracle.lk_flow(fr[0], fr[1], 2, 4, 3)
    finally:
        np_oracle.LK_SPEC_FMA = True
    np.testing.assert_array_equal(rev1.view(np.uint32), g["flow_rev1"].view(np.uint32))
    for f2 in (oracle.lk_flow(fr[0], fr[1], 2, 4, 3), np_oracle.lk_flow(fr[0], fr[1], 2, 4, 3)):
        np.testing.assert_array_equal(f2.view(np.uint32), g["flow_rev2"].view(np.uint32))
    differing = (g["flow_rev1"].view(np.uint32) != g["flow_rev2"].view(np.uint32)).mean()
    assert differing > 0.05, differing                     # the revisions are far enough apart for a drift to show


def test_oracle_reproduces_frontend_golden():
    """cv-decoder's front-end + reduced mode (round 6): the C oracle and the independent NumPy restatement against the committed bytes"""
    from tests import indep_frontend as indep
    g = _load("frontend.npz")
    bgr = g["bgr"]
    gw, gh = (int(v) for v in g["grid_120"])
    assert (gw, gh) == oracle.cv_grid(320, 180, 120, 120) == (120, 67)
    np.testing.assert_array_equal(oracle.resize_linear(bgr[1], gw, gh), g["bgr_small"])
    np.testing.assert_array_equal(indep.resize_linear(bgr[1], gw, gh), g["bgr_small"])
    np.testing.assert_array_equal(oracle.to_gray(bgr[1]), g["gray_full"])
    np.testing.assert_array_equal(indep.to_gray(indep.resize_linear(bgr[1], gw, gh)), g["gray_small"])
    r1, grid, f1 = oracle.cv_decode(bgr[0], bgr[1], oracle.FMT_BGR, process_fullres=False, max_w=120, max_h=120)
    assert grid == (gw, gh) and 0 < len(r1) <= gw * gh
    np.testing.assert_array_equal(r1.view(np.uint32), g["records_pair01"].view(np.uint32))
    np.testing.assert_array_equal(f1.view(np.uint32), g["flow_pair01"].view(np.uint32))
    r2, _, _ = oracle.cv_decode(bgr[1], bgr[2], oracle.FMT_BGR, process_fullres=False, max_w=120, max_h=120, init=f1)
    np.testing.assert_array_equal(r2.view(np.uint32), g["records_pair12_warm"].view(np.uint32))
    r_lk, _, _ = oracle.cv_decode(bgr[0], bgr[1], oracle.FMT_BGR, process_fullres=False, max_w=120, max_h=120, flow="lk")
    np.testing.assert_array_equal(r_lk.view(np.uint32), g["records_pair01_lk"].view(np.uint32))


# ---------------------------------------------------------------- HIP path vs golden (GPU)
@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_hip_sad_matches_golden(ctx):
    g = _load("sad.npz")
    for name, B, R in SAD_KEYS:
        fr = g[f"frames_{name}"]
        ent, best = ctx.sad_flow(fr[0], fr[1], B, R, want_best=True)
        np.testing.assert_array_equal(best, g[f"best_{name}"])
        np.testing.assert_array_equal(ent.view(np.uint32), g[f"entries_{name}"].view(np.uint32))


@pytest.mark.gpu
def test_hip_densify_detect_match_golden(ctx):
    g = _load("densify.npz")
    for k in ("a", "b", "c", "edge"):
        w, h = g[f"wh_{k}"]
        f, cells = ctx.densify(g[f"in_{k}"], int(w), int(h), want_cells=True)
        np.testing.assert_array_equal(cells, g[f"cells_{k}"])
        np.testing.assert_array_equal(f.view(np.uint32), g[f"field_{k}"].view(np.uint32))
    g = _load("detect.npz")
    for k in range(4):
        r = ctx.detect(g[f"in_{k}"])
        assert (r is not None) == bool(g[f"some_{k}"])
        if r:
            assert r[0] == int(g[f"area_{k}"])
            np.testing.assert_array_equal(r[1].view(np.uint32), g[f"field_{k}"].view(np.uint32))


@pytest.mark.gpu
def test_hip_almeida_matches_golden(ctx):
    g = _load("almeida.npz")
    for i in g["field_index"]:
        f = g[f"field_{i}"]
        q, _ = ctx.almeida(f, 1.0, 90.0, use_ransac=False)
        np.testing.assert_allclose(q, g["q_lsq"][i], atol=1e-5, rtol=0)          # north_star tolerance: 1e-4
        q, _ = ctx.almeida(f, 1.0, 90.0, use_ransac=True, num_iters=100, inlier_deg=0.05, num_samples=1000,
                           seed=int(g["ransac_seed0"]) + int(i))
        np.testing.assert_allclose(q, g["q_ransac"][i], atol=1e-4, rtol=0)


@pytest.mark.gpu
def test_hip_flow_is_revision_2_on_the_two_revision_golden(ctx):
    g = _load("flow_rev1.npz")
    fr = g["frames"]
    f = ctx.lk_flow(fr[0], fr[1], 2, 4, 3)
    np.testing.assert_array_equal(f.view(np.uint32), g["flow_rev2"].view(np.uint32))
    assert (f.view(np.uint32) != g["flow_rev1"].view(np.uint32)).any()


@pytest.mark.gpu
def test_hip_flow_decoder_matches_golden(ctx):
    g = _load("flow.npz")
    fr = g["frames"]
    np.testing.assert_array_equal(ctx.lk_flow(fr[0], fr[1], 3, 4, 3).view(np.uint32), g["flow"].view(np.uint32))
    np.testing.assert_array_equal(ctx.contrast_mask(fr[1]), g["mask"])
    rec, _ = ctx.lk_decode(fr[0], fr[1], contrast_mask=True, fullres_records=True)
    np.testing.assert_array_equal(rec.view(np.uint32), g["records"].view(np.uint32))
    ent, (gw, gh) = ctx.lk_decode(fr[0], fr[1], max_w=60, max_h=60, contrast_mask=True)
    assert (gw, gh) == (60, 36)
    np.testing.assert_array_equal(ent.view(np.uint32), g["cells_60x36"].view(np.uint32))


@pytest.mark.gpu
def test_hip_frontend_and_reduced_decoder_match_golden(ctx):
    g = _load("frontend.npz")
    bgr = g["bgr"]
    gw, gh = (int(v) for v in g["grid_120"])
    np.testing.assert_array_equal(ctx.resize_linear(bgr[1], gw, gh, ctx.FMT_BGR), g["bgr_small"])
    np.testing.assert_array_equal(ctx.cv_frontend(bgr[1], ctx.FMT_BGR, False), g["gray_full"])
    np.testing.assert_array_equal(ctx.cv_frontend(bgr[1], ctx.FMT_BGR, True, 120, 120), g["gray_small"])
    kw = dict(max_w=120, max_h=120, contrast_mask=True, reduced=True, fmt=ctx.FMT_BGR)
    ctx.lk_reset()
    assert ctx.lk_push_frame(bgr[0], 5, 6, 3, farneback=True, use_previous=True, **kw) is None
    r1, grid = ctx.lk_push_frame(bgr[1], 5, 6, 3, farneback=True, use_previous=True, **kw)
    r2, _ = ctx.lk_push_frame(bgr[2], 5, 6, 3, farneback=True, use_previous=True, **kw)
    ctx.lk_reset()
    assert grid == (gw, gh)
    np.testing.assert_array_equal(r1.view(np.uint32), g["records_pair01"].view(np.uint32))
    np.testing.assert_array_equal(r2.view(np.uint32), g["records_pair12_warm"].view(np.uint32))
    r_lk, _ = ctx.lk_decode(bgr[0], bgr[1], 3, 4, 3, **kw)
    np.testing.assert_array_equal(r_lk.view(np.uint32), g["records_pair01_lk"].view(np.uint32))


@pytest.mark.gpu
def test_hip_almeida_dense_1080p_matches_golden(ctx):
    """cfg3 at full size: 2,073,600 per-pixel records through the multi-launch LSQ (dense regime: reciprocal-multiply
    quotients, ofps_amd/csrc/almeida.hip:fdiv) vs the oracle's exact-division answer stored in the fixture."""
    g = _load("almeida_dense.npz")
    e = synth.rotation_field(1920, 1080)
    np.testing.assert_array_equal(e[:4], g["first"]); np.testing.assert_array_equal(e[-4:], g["last"])
    q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    np.testing.assert_allclose(q, g["q_lsq"], atol=2e-6, rtol=0)
