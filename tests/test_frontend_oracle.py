"""cv-decoder's frame front-end (cv-decoder/src/lib.rs:98-135), CPU side: the C oracle against an independent NumPy restatement, hand-derived
values and the properties OpenCV's definitions imply.  PARITY UNPINNED (OpenCV is absent): tools/external_parity carries the vectors for a
machine that has cv2."""
import numpy as np
import pytest

import oracle
from tests import indep_frontend as indep


def _img(H, W, cn, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (H, W, cn), dtype=np.uint8)
    return a[:, :, 0].copy() if cn == 1 else a


def test_cv_grid_is_the_reference_arithmetic():
    # cv-decoder/src/lib.rs:98-121 by hand: 1920x1080 under (150, 150): width_based (150, 150*1080/1920 = 84), height_based (150*1920/1080 = 266, 150)
    assert oracle.cv_grid(1920, 1080) == (150, 84)
    assert oracle.cv_grid(3840, 2160) == (150, 84)
    assert oracle.cv_grid(1080, 1920) == (84, 150)              # portrait: height_based (150*1080/1920 = 84, 150)
    assert oracle.cv_grid(96, 64) == (96, 64)                   # the cap is the frame itself
    assert oracle.cv_grid(100, 300) == (50, 150)
    assert oracle.cv_grid(640, 360, 2000, 2000) == (640, 360)
    assert oracle.cv_grid(1920, 1080, 2000, 150) == (266, 150)  # height_based: 150 * 1920 / 1080 = 266 (integer division)
    for W, H, mw, mh in [(1920, 1080, 150, 150), (97, 61, 40, 150), (5, 1000, 150, 150), (1000, 5, 3, 2)]:
        assert oracle.cv_grid(W, H, mw, mh) == indep.cv_grid(W, H, mw, mh)


def test_resize_coefficient_tables():
    """half-pixel centres, the horizontal edge rule, 11-bit coefficients rounded half to even"""
    ofs, coef = oracle.resize_linear_axis(1920, 150, True)
    # d = 0: (0.5 * 12.8 - 0.5) = 5.9 -> s 5, f 0.9 -> (205, 1843);  d = 2: 31.5 -> (1024, 1024)
    assert ofs[0] == 5 and tuple(coef[0]) == (205, 1843)
    assert ofs[2] == 31 and tuple(coef[2]) == (1024, 1024)
    assert (coef.sum(1) >= 2047).all() and (coef.sum(1) <= 2049).all()
    # enlarging: the first / last columns fall outside: the edge rule pins them to the border pixel with weight 2048
    ofs, coef = oracle.resize_linear_axis(4, 8, True)
    assert ofs[0] == 0 and tuple(coef[0]) == (2048, 0)          # f = -0.25 -> s = -1 -> (0, 0)
    assert ofs[-1] == 3 and tuple(coef[-1]) == (2048, 0)
    # the vertical axis has no edge rule: s = -1 survives (the rows are clamped instead)
    ofs, coef = oracle.resize_linear_axis(4, 8, False)
    assert ofs[0] == -1 and tuple(coef[0]) == (512, 1536)
    for src, dst in [(1920, 150), (1080, 84), (7, 5), (5, 7), (333, 77), (64, 63)]:
        for edge in (True, False):
            o, c = oracle.resize_linear_axis(src, dst, edge)
            s, c0, c1 = indep._axis_table(src, dst, edge)
            np.testing.assert_array_equal(o, s)
            np.testing.assert_array_equal(c[:, 0], c0)
            np.testing.assert_array_equal(c[:, 1], c1)


@pytest.mark.parametrize("W,H,dw,dh,cn", [(1920, 1080, 150, 84, 3), (1920, 1080, 150, 84, 1), (333, 77, 150, 34, 3), (97, 61, 40, 25, 4),
                                           (64, 48, 64, 48, 3), (300, 168, 150, 84, 3), (300, 168, 150, 84, 1), (8, 6, 3, 5, 1), (5, 7, 9, 11, 3),
                                           (2, 2, 1, 1, 1), (1, 1, 1, 1, 3), (640, 360, 150, 84, 4), (3, 1000, 1, 150, 1)])
def test_resize_matches_the_independent_restatement(W, H, dw, dh, cn):
    a = _img(H, W, cn, W * 7 + H + cn)
    np.testing.assert_array_equal(oracle.resize_linear(a, dw, dh), indep.resize_linear(a, dw, dh))


def test_resize_hand_values_and_properties():
    # 4 x 1 -> 2 x 1 is the exact-2x rule only when BOTH axes halve; here the general path: columns at 0.5 and 2.5 -> (a + b) / 2 with 1024 / 1024
    row = np.array([[10, 20, 30, 40]], np.uint8)
    out = oracle.resize_linear(row, 2, 1)
    # D = 10 * 1024 + 20 * 1024 = 30720; vertical: s = 0, f = 0 -> b = (2048, 0): ((2048 * (30720 >> 4)) >> 16) = 60; (60 + 0 + 2) >> 2 = 15
    assert out.tolist() == [[15, 35]]
    # exact 2 x 2 reduction = area mean with rounding
    blk = np.array([[1, 2], [3, 5]], np.uint8)
    assert oracle.resize_linear(blk, 1, 1).tolist() == [[(1 + 2 + 3 + 5 + 2) >> 2]]
    # a constant image stays constant (coefficient pairs that sum to 2047 would lose a level: they do not occur at these sizes)
    for v in (0, 1, 127, 255):
        c = np.full((1080, 1920), v, np.uint8)
        assert (oracle.resize_linear(c, 150, 84) == v).all()
    # same size: a copy
    a = _img(48, 64, 3, 1)
    np.testing.assert_array_equal(oracle.resize_linear(a, 64, 48), a)
    # channels are independent
    r = oracle.resize_linear(a, 21, 17)
    for c in range(3):
        np.testing.assert_array_equal(r[:, :, c], oracle.resize_linear(a[:, :, c].copy(), 21, 17))
    # the two published vertical forms differ by at most one level (the kit tells which one a given OpenCV build runs)
    b = _img(1080, 1920, 1, 9)
    d = oracle.resize_linear(b, 150, 84).astype(int) - oracle.resize_linear(b, 150, 84, variant=1).astype(int)
    assert np.abs(d).max() <= 1


def test_gray_formula():
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 90]]], np.uint8)      # BGR
    want = [255, 0, (255 * 1868 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 4899 + 8192) >> 14, (10 * 1868 + 200 * 9617 + 90 * 4899 + 8192) >> 14]
    assert want[2:5] == [29, 150, 76]
    assert oracle.to_gray(px, oracle.FMT_BGR)[0].tolist() == want
    rgba = np.concatenate([px[:, :, ::-1], np.full((1, 6, 1), 7, np.uint8)], 2)
    assert oracle.to_gray(rgba, oracle.FMT_RGBA)[0].tolist() == want
    bgra = np.concatenate([px, np.full((1, 6, 1), 200, np.uint8)], 2)
    assert oracle.to_gray(bgra, oracle.FMT_BGRA)[0].tolist() == want
    a = _img(77, 333, 3, 5)
    np.testing.assert_array_equal(oracle.to_gray(a), indep.to_gray(a, "bgr"))
    a4 = _img(77, 333, 4, 6)
    np.testing.assert_array_equal(oracle.to_gray(a4, oracle.FMT_RGBA), indep.to_gray(a4, "rgba"))


def test_cv_frontend_order_is_resize_then_gray():
    """cv-decoder/src/lib.rs:124-135: the COLOUR frame is resized, then converted -- not the other way round (the results differ)"""
    a = _img(360, 640, 3, 11)
    g = oracle.cv_frontend(a, oracle.FMT_BGR, process_fullres=False)
    assert g.shape == (84, 150)
    np.testing.assert_array_equal(g, oracle.to_gray(oracle.resize_linear(a, 150, 84)))
    other = oracle.resize_linear(oracle.to_gray(a), 150, 84)
    assert (g != other).any() and np.abs(g.astype(int) - other.astype(int)).max() <= 2
    np.testing.assert_array_equal(oracle.cv_frontend(a, oracle.FMT_BGR), oracle.to_gray(a))
    y = a[:, :, 0].copy()
    np.testing.assert_array_equal(oracle.cv_frontend(y), y)


def test_cv_decode_reduced_record_lattice():
    """ "Process Fullres" = false (cv-decoder/src/lib.rs:239-243,274-276): one record per unmasked pixel of the REDUCED frame at
    ((x + .5) / gw, (y + .5) / gh), motion = flow / (gw, gh); ~12.6 k records at the default cap, not W * H"""
    from ofps_amd import synth
    fr = synth.luma_sequence(2, 640, 360, max_step=6, seed=4)
    rec, (gw, gh), flow = oracle.cv_decode(fr[0], fr[1], process_fullres=False, contrast_mask_on=False)
    assert (gw, gh) == (150, 84) and flow.shape == (84, 150, 2) and len(rec) == 150 * 84
    yy, xx = np.meshgrid(np.arange(84), np.arange(150), indexing="ij")
    np.testing.assert_array_equal(rec[:, 0], ((xx.ravel() + np.float32(0.5)) * (np.float32(1) / np.float32(150))).astype(np.float32))
    np.testing.assert_array_equal(rec[:, 1], ((yy.ravel() + np.float32(0.5)) * (np.float32(1) / np.float32(84))).astype(np.float32))
    np.testing.assert_array_equal(rec[:, 2], flow[..., 0].ravel() * (np.float32(1) / np.float32(150)))
    rec_m, _, _ = oracle.cv_decode(fr[0], fr[1], process_fullres=False)
    assert 0 < len(rec_m) <= 150 * 84
    # the default mode reaches the same grid through the densifier
    rec_f, grid_f, flow_f = oracle.cv_decode(fr[0], fr[1], process_fullres=True)
    assert grid_f == (150, 84) and flow_f.shape == (360, 640, 2) and len(rec_f) <= 150 * 84
