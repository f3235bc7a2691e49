"""CPU tests that pin the oracle: the reference's own known-answer tests, the camera doctest, and a
cross-check of the C restatement against the independent NumPy restatement."""
import numpy as np
import pytest

import oracle
from oracle import np_oracle as npo
from ofps_amd import synth

import almeida_cases as ac


# ---- reference tests (almeida-estimator/src/lib.rs:359-372, ofps/src/camera.rs:139-149) ----
def test_reference_point_angle_doctest():
    cam = oracle.camera(1.0, 90.0)
    ang = oracle.camera_point_angle(cam, [1.0, 0.5])
    assert abs(np.degrees(ang[0]) - 45.0) < 0.01


def test_reference_rotation_default():
    """test_rotation_default: LSQ, 32 rotations, error < 10 % of the rotation."""
    cam = oracle.camera(1.0, 90.0)
    worst = 0.0
    for rot, ang, q, field in ac.cases():
        est = oracle.solve_ypr_given(field, cam)
        err = ac.error_deg(q, est)
        assert err < 0.1 * rot or err == 0.0, (rot, ang, err)
        worst = max(worst, err / (0.1 * rot))
    assert worst < 0.05          # far inside the reference's bound


def test_reference_rotation_ransac():
    """test_rotation_ransac: 100 iterations; the build's seeded sampler stands in for thread_rng."""
    cam = oracle.camera(1.0, 90.0)
    for i, (rot, ang, q, field) in enumerate(ac.cases()):
        est = oracle.solve_ypr_ransac(field, cam, 100, 0.05, 1000, seed=1234 + i)
        err = ac.error_deg(q, est)
        assert err < 0.1 * rot or err == 0.0, (rot, ang, err)


# ---- C restatement vs the NumPy restatement ----
def _entries(n, seed, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    e = np.empty((n, 4), np.float32)
    e[:, :2] = rng.uniform(lo, hi, (n, 2)).astype(np.float32)
    e[:, 2:] = rng.normal(0, 0.01, (n, 2)).astype(np.float32)
    return e


@pytest.mark.parametrize("n,w,h", [(1000, 14, 14), (1000, 150, 84), (300, 7, 3), (0, 4, 4)])
def test_densify_c_vs_numpy(n, w, h):
    e = _entries(n, 10 + n + w)
    f_c, cells_c = oracle.densify(e, w, h, want_cells=True)
    f_n, cells_n = npo.densify(e, w, h)
    np.testing.assert_array_equal(cells_c.astype(np.int64), cells_n.reshape(-1, 2))
    np.testing.assert_array_equal(f_c.view(np.uint32), f_n.view(np.uint32))


def test_densify_semantics():
    # empty cell: 0/eps = 0; one vector: divided by 1+eps, not 1
    e = np.array([[0.5, 0.5, 0.25, -0.5]], np.float32)
    f, cnt = oracle.densify(e, 3, 3, want_counts=True)
    assert (f[0, 0] == 0).all()
    one_eps = np.float32(1.0) + np.float32(1.1920929e-07)
    assert cnt[1, 1, 0] == one_eps
    assert f[1, 1, 0] == np.float32(0.25) / one_eps
    # two vectors: eps is absorbed (2 + eps rounds to 2)
    e2 = np.array([[0.5, 0.5, 0.25, 0], [0.5, 0.5, 0.5, 0]], np.float32)
    f2, cnt2 = oracle.densify(e2, 3, 3, want_counts=True)
    assert cnt2[1, 1, 0] == np.float32(2.0)
    # rounding is half away from zero: pos*(w-1) = 0.5 -> cell 1
    e3 = np.array([[0.25, 0.75, 1, 1]], np.float32)
    _, cells = oracle.densify(e3, 3, 3, want_cells=True)
    assert tuple(cells[0]) == (1, 2)
    # clamp quirk (unverified vs rustc): one coordinate out of range moves both
    e4 = np.array([[-0.1, 0.7, 1, 1], [0.3, 1.0, 1, 1], [np.nan, 0.2, 1, 1]], np.float32)
    _, cells = oracle.densify(e4, 5, 5, want_cells=True)
    assert [tuple(c) for c in cells] == [(0, 0), (4, 4), (0, 0)]


def test_densify_to_entries_order_and_positions():
    e = _entries(200, 3)
    out = oracle.densify_to_entries(e, 6, 4)
    f, cells = oracle.densify(e, 6, 4, want_cells=True)
    visited = sorted({(int(x), int(y)) for x, y in cells})          # BTreeSet<(x,y)> order
    assert len(out) == len(visited)
    for row, (x, y) in zip(out, visited):
        assert row[0] == (np.float32(x) + np.float32(0.5)) * (np.float32(1) / np.float32(6))
        assert row[1] == (np.float32(y) + np.float32(0.5)) * (np.float32(1) / np.float32(4))
        assert (row[2:] == f[y, x]).all()


@pytest.mark.parametrize("seed,scale", [(0, 0.3), (1, 30.0), (2, 1.0), (3, 0.7)])
def test_detect_c_vs_numpy(seed, scale):
    e = _entries(4000, 50 + seed)
    e[:, 2:] *= np.float32(scale)
    r_c = oracle.detect_motion(e)
    r_n = npo.detect_motion(e)
    assert (r_c is None) == (r_n is None)
    if r_c is not None:
        assert r_c[0] == r_n[0]
        np.testing.assert_array_equal(r_c[1].view(np.uint32), r_n[1].view(np.uint32))


def test_detect_block_dim():
    assert oracle.block_dim(0.05, 3) == 14 == npo.block_dim(0.05, 3)       # defaults
    assert oracle.block_dim(0.01, 16) == 160
    assert oracle.block_dim(1.0, 1) == 1


def test_camera_delta_c_vs_numpy_closed_form():
    cam_c = oracle.camera(16 / 9, 22.275); cam_n = npo.Camera(16 / 9, 22.275)
    e = _entries(500, 4)
    R = npo.rot3_from_euler(0.01, -0.02, 0.005)
    R4 = np.eye(4, dtype=np.float32); R4[:3, :3] = R
    d_c = np.array([oracle.camera_delta(cam_c, p, R4) for p in e[:, :2]])
    d_n = cam_n.delta(e[:, :2], R)
    np.testing.assert_array_equal(d_c.view(np.uint32), d_n.view(np.uint32))   # closed form == generic path, bit for bit


def test_solve_ypr_c_vs_numpy():
    cam_c = oracle.camera(16 / 9, 22.275); cam_n = npo.Camera(16 / 9, 22.275)
    e = synth.rotation_field(48, 27)
    q_c = oracle.solve_ypr_given(e, cam_c)
    q_n = npo.solve_ypr_given(e, cam_n)
    np.testing.assert_allclose(q_c, q_n, atol=1e-6, rtol=0)


def test_lu3_solve_matches_numpy_and_singular():
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.normal(size=(3, 3)).astype(np.float32); b = rng.normal(size=3).astype(np.float32)
        x = oracle.lu3_solve(a, b)
        np.testing.assert_allclose(x, np.linalg.solve(a.astype(np.float64), b), rtol=2e-3, atol=2e-4)
    assert oracle.lu3_solve(np.zeros((3, 3), np.float32), np.ones(3, np.float32)) is None   # -> zero step (lib.rs:183)


def test_sampler_is_a_permutation():
    for n in (1, 2, 3, 10, 1000, 2500, 129600):
        idx = [oracle.sample_index(77, 5, 1, i, n) for i in range(min(n, 3000))]
        assert len(set(idx)) == len(idx) and max(idx) < n
    a = [oracle.sample_index(1, 0, 0, i, 1000) for i in range(3)]
    b = [oracle.sample_index(1, 1, 0, i, 1000) for i in range(3)]
    assert a != b


def test_sad_c_vs_numpy_small():
    fr = synth.luma_sequence(2, 96, 64, max_step=8)
    _, best_c = oracle.sad_flow(fr[0], fr[1], 16, 8)
    np.testing.assert_array_equal(best_c, npo.sad_flow(fr[0], fr[1], 16, 8))
    flat = np.full((2, 48, 64), 9, np.uint8)
    _, best = oracle.sad_flow(flat[0], flat[1], 16, 8)
    assert (best == 0).all()                       # all-tie -> smallest d2 -> (0,0)


def test_sad_openmp_equals_scalar():
    fr = synth.luma_sequence(2, 320, 192, max_step=16)
    e1, b1 = oracle.sad_flow(fr[0], fr[1], 16, 16, threads=1, simd=False)      # the definition loop
    e2, b2 = oracle.sad_flow(fr[0], fr[1], 16, 16, threads=max(2, oracle.num_threads()))
    np.testing.assert_array_equal(b1, b2)
    np.testing.assert_array_equal(e1.view(np.uint32), e2.view(np.uint32))


def test_almeida_all_core_forms_equal_the_single_thread_oracle_bit_for_bit():
    """bench.py's all-core CPU leg times orc_solve_ypr_given_mt / orc_solve_ypr_ransac_mt: the
    per-vector loops and the hypotheses are spread over OpenMP threads, every sum keeps its sequential order, the first
    maximum wins -- so they are the single-thread restatement bit for bit, whatever the thread count."""
    cam = oracle.camera(16 / 9, 22.275)
    for shape, frac in (((120, 67), 0.2), ((64, 36), 0.0), ((7, 3), 0.0)):
        e = synth.rotation_field(*shape, outlier_frac=frac)
        q1 = oracle.solve_ypr_given(e, cam)
        r1 = oracle.solve_ypr_ransac(e, cam, 50, 0.05, 1000, seed=9)
        for t in (2, 3, 8):
            np.testing.assert_array_equal(oracle.solve_ypr_given(e, cam, threads=t).view(np.uint32), q1.view(np.uint32))
            np.testing.assert_array_equal(oracle.solve_ypr_ransac(e, cam, 50, 0.05, 1000, seed=9, threads=t).view(np.uint32),
                                          r1.view(np.uint32))
    # fewer than 3 inliers -> identity, in both forms (almeida-estimator/src/lib.rs:246-250)
    e = synth.rotation_field(2, 1)
    np.testing.assert_array_equal(oracle.solve_ypr_ransac(e, cam, 10, 0.05, 1000, seed=1, threads=4),
                                  oracle.solve_ypr_ransac(e, cam, 10, 0.05, 1000, seed=1))


def test_numpy_fma32_is_correctly_rounded():
    """The NumPy restatement of the LK spec (revision 2) needs an f32 fused multiply-add without hardware help: exact f64
    product, round-to-odd f64 sum, one final rounding.  Checked against libm's fmaf on random operands with heavy
    cancellation and on the double-rounding trap: exact results a hair below / above a tie between two f32 neighbours,
    where rounding through plain f64 picks the wrong one."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6"); libm.fmaf.restype = ctypes.c_float; libm.fmaf.argtypes = [ctypes.c_float] * 3
    rng = np.random.default_rng(3)
    n = 20000
    a = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)).astype(np.float32)
    c = (-a.astype(np.float64) * b.astype(np.float64) * (1 + rng.uniform(-1e-6, 1e-6, n))).astype(np.float32)
    c[::3] = (rng.standard_normal(len(c[::3])) * 10 ** rng.uniform(-3, 3, len(c[::3]))).astype(np.float32)
    # the trap: a * b = 2^-24 (1 - 2^-46 k^2) or 2^-24 (1 + 2^-46 k^2 + ...), c = r with an odd or even last bit
    ta, tb, tc = [], [], []
    for k in range(1, 6):
        for r in (1.0 + 2.0 ** -23, 1.0 + 2.0 ** -22, 1.5, 1.0 + 3 * 2.0 ** -23):
            for sg in (1.0, -1.0):
                for sb in (1.0, -1.0):
                    ta.append(sg * 2.0 ** -12 * (1 + k * 2.0 ** -23)); tb.append(2.0 ** -12 * (1 + sb * k * 2.0 ** -23)); tc.append(sg * r)
    a = np.concatenate([a, np.array(ta, np.float32)]); b = np.concatenate([b, np.array(tb, np.float32)])
    c = np.concatenate([c, np.array(tc, np.float32)])
    want = np.array([libm.fmaf(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], np.float32)
    np.testing.assert_array_equal(npo.fma32(a, b, c).view(np.uint32), want.view(np.uint32))
    naive = (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    assert (naive.view(np.uint32) != want.view(np.uint32)).any(), "the trap cases should defeat rounding through plain f64"


def test_lk_spec_revision_is_the_fused_one_in_both_restatements():
    assert oracle.lk_spec_revision() == 2 and npo.LK_SPEC_FMA is True


def test_interpolate_empty_cells_fills_everything():
    e = _entries(40, 8)
    f = oracle.densify_interpolated(e, 12, 9)
    plain = oracle.densify(e, 12, 9)
    filled = (plain != 0).any(-1)
    np.testing.assert_array_equal(f[filled].view(np.uint32) != 0, plain[filled].view(np.uint32) != 0)
    assert np.isfinite(f).all()
    # no vectors at all: stays empty (motion_field.rs:243-246)
    assert (oracle.densify_interpolated(np.zeros((0, 4), np.float32), 5, 5) == 0).all()


def test_lk_flow_oracle_recovers_translation_and_zero():
    base = synth.luma_sequence(1, 256 + 64, 160 + 64, max_step=0, noise=0, seed=5)[0]
    prev = np.ascontiguousarray(base[32:32 + 160, 32:32 + 256])
    for dx, dy in [(3, -2), (0, 0)]:
        cur = np.ascontiguousarray(base[32 - dy:32 - dy + 160, 32 - dx:32 - dx + 256])
        f = oracle.lk_flow(prev, cur, 3, 4, 3)
        assert np.abs(f[24:-24, 24:-24] - np.array([dx, dy], np.float32)).mean() < 1e-3
    e = oracle.flow_to_entries(np.ones((4, 6, 2), np.float32))
    assert e.shape == (24, 4) and e[0, 0] == np.float32(0.5) * (np.float32(1) / np.float32(6)) and e[7, 1] == np.float32(1.5) * np.float32(0.25)
    with pytest.raises(ValueError):
        oracle.lk_flow(prev, prev, 0, 4, 3)


def test_contrast_mask_known_answers():
    """cv-decoder/src/lib.rs:203-237.  An impulse of 6 on black gives Sobel(1,1,k5) = 6*K[i]*K[j] with K = [-1,-2,0,2,1]:
    only the two +4 taps exceed 20, at (y0+1, x0+1) and (y0-1, x0-1); the mask is the union of two copies of
    OpenCV's 11x11 MORPH_ELLIPSE element (the matrix getStructuringElement documents) centred there."""
    ell = np.array([[0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0],
                    [0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0],
                    [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0],
                    [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                    [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                    [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                    [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                    [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                    [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0],
                    [0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0],
                    [0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]], np.uint8)
    g = np.zeros((40, 50), np.uint8)
    g[20, 25] = 6
    want = np.zeros_like(g)
    for cy, cx in ((21, 26), (19, 24)):
        want[cy - 5:cy + 6, cx - 5:cx + 6] |= ell
    np.testing.assert_array_equal(oracle.contrast_mask(g), want)
    np.testing.assert_array_equal(npo.contrast_mask(g), want)
    g[20, 25] = 5                                           # 5*4 = 20 is not > 20
    assert not oracle.contrast_mask(g).any()
    # negative mixed derivative never passes (THRESH_BINARY on the signed response)
    g[:] = 0; g[20, 25] = 255
    m = oracle.contrast_mask(g)
    assert m[20 + 1, 25 + 1] and m[20 - 1, 25 - 1]
    # an image corner: out-of-image taps of the dilation are ignored, Sobel reflects (101) at the border
    r = np.random.default_rng(3).integers(0, 256, (9, 12), dtype=np.uint8)
    np.testing.assert_array_equal(oracle.contrast_mask(r), npo.contrast_mask(r))


def test_masked_records_keep_raster_order():
    flow = np.random.default_rng(1).normal(0, 1, (6, 8, 2)).astype(np.float32)
    mask = (np.random.default_rng(2).random((6, 8)) < 0.5).astype(np.uint8)
    full = oracle.flow_to_entries(flow)
    np.testing.assert_array_equal(oracle.masked_flow_to_entries(flow, mask), full[mask.reshape(-1) != 0])
    np.testing.assert_array_equal(oracle.masked_flow_to_entries(flow, None), full)


@pytest.mark.parametrize("W,H,levels,radius,iters", [(64, 48, 3, 4, 3), (97, 61, 2, 2, 2), (40, 30, 1, 6, 2), (33, 17, 3, 1, 4)])
def test_lk_flow_c_vs_numpy_bit_exact(W, H, levels, radius, iters):
    """N2 is build-defined (the reference calls OpenCV): two independent restatements of the spec must agree bit for bit,
    so a self-consistent-but-wrong C oracle cannot hide behind the GPU parity tests."""
    fr = synth.luma_sequence(2, W, H, max_step=2, seed=W + H)
    a = npo.lk_flow(fr[0], fr[1], levels, radius, iters)
    b = oracle.lk_flow(fr[0], fr[1], levels, radius, iters)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,w,h,seed", [(40, 12, 9, 8), (5, 16, 16, 1), (300, 40, 23, 2), (1, 7, 5, 3), (0, 5, 5, 4), (60, 14, 14, 5)])
def test_interpolate_empty_cells_c_vs_python_bit_exact(n, w, h, seed):
    """interpolate_empty_cells (ofps/src/motion_field.rs:193-294) has no test in the reference: the C oracle and a second
    restatement written from the reference's text (ordered set = its BTreeSet) must agree bit for bit."""
    e = _entries(n, seed)
    a = npo.densify_interpolated(e, w, h)
    b = oracle.densify_interpolated(e, w, h)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
