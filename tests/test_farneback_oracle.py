"""CPU tests of the Farneback restatement (oracle/farneback_oracle.c): against an INDEPENDENT float64 scipy restatement (tests/indep_farneback.py),
against the committed vectors of the external-check kit, and against a planted homography.  PARITY UNPINNED with respect to the reference:
cv-decoder calls OpenCV (cv-decoder/src/lib.rs:188-199), which is neither under /root/reference nor installed;
tools/external_parity/opencv_compare.py is the check for anyone who has it."""
import os

import numpy as np
import pytest

import indep_farneback as F
import oracle
from ofps_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layers_and_kernels_are_opencvs():
    # calcOpticalFlowFarneback(levels = 5) on 1080p: six layers, sizes by cvRound (round half to even): 1080 / 32 = 33.75 -> 34, 1080 / 16 = 67.5 -> 68
    assert oracle.farneback_layers(1920, 1080, 5) == [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68), (60, 34)]
    assert oracle.farneback_layers(640, 360, 5) == [(640, 360), (320, 180), (160, 90), (80, 45)]        # 40 x 22.5 is under 32 px: the pyramid ends
    assert oracle.farneback_layers(40, 33, 5) == [(40, 33)]
    # blur of layer k: sigma = (2^k - 1) / 2, ksize = max(cvRound(5 sigma) | 1, 3): 3, 3, 9, 19, 39, 79 taps; [1 2 1] / 4 at k = 0
    assert [len(oracle.farneback_kernels(k)[0]) for k in range(6)] == [3, 3, 9, 19, 39, 79]
    np.testing.assert_array_equal(oracle.farneback_kernels(0)[0], np.array([0.25, 0.5, 0.25], np.float32))
    for k in range(1, 6):
        t = oracle.farneback_kernels(k)[0]
        assert abs(float(t.sum()) - 1) < 1e-6 and np.array_equal(t, t[::-1])
        np.testing.assert_allclose(t, F.blur_taps(k), rtol=0, atol=1e-9)
    # expansion kernel: radius poly_n = 7, sigma 1.5; the four inverse-moment entries against a numerical inverse of the 6 x 6 matrix
    _, g, xg, xxg, ig = oracle.farneback_kernels(0, 7, 1.5)
    gi, xgi, xxgi, igi = F.poly_kernels(7, 1.5)
    np.testing.assert_allclose(g, gi[7:], atol=1e-9); np.testing.assert_allclose(xg, xgi[7:], atol=1e-8); np.testing.assert_allclose(xxg, xxgi[7:], atol=1e-7)
    np.testing.assert_allclose(ig, igi, rtol=1e-9)


@pytest.mark.parametrize("k", [0, 1, 3])
def test_layer_image_and_expansion_stage_by_stage(k):
    fr, _ = synth.rotation_clip([(0.1, 0.05, 0.2)], 320, 180, 60.0, seed=3)
    I, R = oracle.farneback_layer(fr[0], k)
    h, w = I.shape
    Ii = F.layer_image(fr[0], k, w, h)
    assert np.abs(I - Ii).max() < 1e-4                      # f32 accumulation of up to 19 x 2 taps on values up to 255
    assert np.abs(R - F.poly_exp(Ii)).max() < 1e-4


@pytest.mark.parametrize("content", ["camera", "regions"])
def test_flow_equals_the_independent_restatement(content):
    if content == "camera":
        fr, _ = synth.rotation_clip([(0.15, -0.1, 0.3)], 320, 180, 60.0, seed=9)
    else:
        fr = synth.luma_sequence(2, 322, 181, max_step=3, seed=31)
    f_o = oracle.farneback_flow(fr[0], fr[1])
    f_i = F.farneback(fr[0], fr[1])
    assert np.abs(f_o - f_i).max() < 1e-4, float(np.abs(f_o - f_i).max())          # measured ~5e-6: f32 vs f64 rounding only
    f_o = oracle.farneback_flow(fr[0], fr[1], levels=2, winsize=9, iters=2, poly_n=5, poly_sigma=1.1)
    f_i = F.farneback(fr[0], fr[1], levels=2, winsize=9, iters=2, poly_n=5, poly_sigma=1.1)
    assert np.abs(f_o - f_i).max() < 1e-4


def test_planted_homography_is_recovered():
    W, H = 480, 270
    e = (0.1, -0.05, 0.15)
    fr, _ = synth.rotation_clip([e], W, H, 60.0, seed=4)
    f = oracle.farneback_flow(fr[0], fr[1])
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pos = np.stack([(xx.ravel() + .5) / W, (yy.ravel() + .5) / H], 1)
    truth = synth.rotation_delta(pos, W / H, 60.0, synth.euler_rot3(*np.radians(e)).T).reshape(H, W, 2) * [W, H]
    err = np.linalg.norm(f - truth, axis=2)
    assert np.median(err) < 0.05 and np.percentile(err[20:-20, 20:-20], 99) < 0.3


def test_committed_vectors_of_the_external_kit_are_this_oracles_output():
    """every expected array of tools/external_parity/data/farneback_pairs.npz -- cv-decoder's call cold and warm, the two other published forms
    of the separable Gaussian, the stage ablations, the layer images, the front-end -- is what the oracle computes today"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_farneback_pairs", os.path.join(ROOT, "tools", "external_parity", "make_farneback_pairs.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    d = np.load(gen.PATH)
    seen = set()
    for name in ("camera", "regions"):
        exp = gen.expected(d[name + "_prev"], d[name + "_cur"], name)
        for k, v in exp.items():
            np.testing.assert_array_equal(v.view(np.uint8), d[k].view(np.uint8), err_msg=k)
            seen.add(k)
        # the other forms of the Gaussian move the flow by a few 1e-7 px: far inside north_star's 1e-4, and not the same bits
        for v in (1, 2):
            dd = np.abs(exp[f"{name}_flow_v{v}"] - exp[name + "_flow"])
            assert 0 < dd.max() < 2e-5, (name, v, float(dd.max()))
        np.testing.assert_array_equal(exp[name + "_layer1"], exp[name + "_layer1_v1"])            # 3 taps: SymmRowSmallFilter in every form
        assert (exp[name + "_layer2"] != exp[name + "_layer2_v1"]).any()                         # 9 taps: the row pass differs
    assert seen | {n + s for n in ("camera", "regions") for s in ("_prev", "_cur")} == set(d.files)


def test_initial_flow_and_bad_arguments():
    fr = synth.luma_sequence(3, 160, 96, max_step=2, seed=8)
    first = oracle.farneback_flow(fr[0], fr[1])
    warm = oracle.farneback_flow(fr[1], fr[2], init=first)
    cold = oracle.farneback_flow(fr[1], fr[2])
    assert warm.shape == cold.shape and np.isfinite(warm).all() and not np.array_equal(warm, cold)
    assert np.abs(warm - cold).mean() < 0.2
    for kw in (dict(winsize=12), dict(iters=0), dict(poly_n=0)):
        with pytest.raises(ValueError):
            oracle.farneback_flow(fr[0], fr[1], **kw)
