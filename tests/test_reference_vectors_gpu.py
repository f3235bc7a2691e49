"""-m gpu: the HIP path against pins that do not come from the oracle (tests/test_reference_vectors.py explains each),
the dense Almeida regime at the reference test's own camera, and BASELINE configs[2] / configs[4] at full size."""
import numpy as np
import pytest

import oracle
from ofps_amd import synth
import indep_model as im
import almeida_cases as ac
import test_reference_vectors as rv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


# ---- hand-derived literals through the C ABI ------------------------------------------------------------------------
def test_hip_densifier_hand_derived_case(ctx):
    field, cells = ctx.densify(rv.DENSIFY_ENTRIES, 3, 3, want_cells=True)
    assert [tuple(int(v) for v in c) for c in cells] == rv.DENSIFY_CELLS
    np.testing.assert_array_equal(field.view(np.uint32), rv.densify_expected_field())


@pytest.mark.parametrize("name", sorted(rv.DETECT_CASES))
def test_hip_detector_hand_derived_cases(ctx, name):
    ent, want = rv.DETECT_CASES[name]
    got = ctx.detect(ent)
    if want is None:
        assert got is None
        return
    assert got is not None and got[0] == want[0]
    np.testing.assert_array_equal(got[1].view(np.uint32), want[1])


# ---- reference-held input: docs/report/mfield/base.csv --------------------------------------------------------------
@pytest.mark.parametrize("ransac", [False, True])
def test_hip_almeida_on_the_reference_table(ctx, ransac):
    base = rv.load("base").astype(np.float32)
    q, _ = ctx.almeida(base, *rv.CAM, use_ransac=ransac, num_iters=100, inlier_deg=0.05, num_samples=1000, seed=5)
    want = im.quat_wijk(rv._model_quat(np.radians([0.0, -20.0, 2.0]), "pry").inv()).astype(np.float32)
    assert np.degrees(oracle.quat_angle_to(want, q)) < 0.01          # planted rotation of the table, 0.05 % of 20 degrees
    cam = oracle.camera(*rv.CAM)
    q_o = oracle.solve_ypr_ransac(base, cam, 100, 0.05, 1000, seed=5) if ransac else oracle.solve_ypr_given(base, cam)
    np.testing.assert_allclose(q, q_o, atol=1e-4 if ransac else 2e-6, rtol=0)


# ---- the reference's known-answer test on fields built by the independent model ------------------------------------
def test_hip_almeida_known_answer_on_independent_fields(ctx):
    cam = oracle.camera(1.0, 90.0)
    for rot in ac.ROTS:
        for (r, p, y) in ac.angle_combos(rot):
            q_i, ent, keep = im.almeida_test_field(r, p, y)
            e = ent[keep].astype(np.float32)
            est, _ = ctx.almeida(e, 1.0, 90.0, use_ransac=False)
            err = np.degrees(oracle.quat_angle_to(im.quat_wijk(q_i).astype(np.float32), est))
            assert err < 0.1 * rot or err < 1e-4, (rot, (r, p, y), err)              # almeida-estimator/src/lib.rs:343-348
            np.testing.assert_allclose(est, oracle.solve_ypr_given(e, cam), atol=1e-5, rtol=0)


# ---- dense regime (N > 65,536: reciprocal-multiply quotients) at the reference test's camera and rotations ----------
@pytest.mark.parametrize("rot", [0.1, 1.0, 10.0])
def test_hip_almeida_dense_regime_reference_camera(ctx, rot):
    """get_grid up-sampled to 300x300 (almeida-estimator/src/lib.rs:308-348 uses 50x50): ~70,000 kept vectors."""
    cam = oracle.camera(1.0, 90.0)
    for (r, p, y) in ac.angle_combos(rot)[1:]:
        q_i, ent, keep = im.almeida_test_field(r, p, y, n=300)
        e = ent[keep].astype(np.float32)
        assert len(e) > 65536
        est, _ = ctx.almeida(e, 1.0, 90.0, use_ransac=False)
        err = np.degrees(oracle.quat_angle_to(im.quat_wijk(q_i).astype(np.float32), est))
        assert err < 0.1 * rot, (rot, (r, p, y), err)
        np.testing.assert_allclose(est, oracle.solve_ypr_given(e, cam), atol=2e-6, rtol=0)


@pytest.mark.parametrize("cam_rot", [((1.0, 90.0), (10.0, 10.0, 10.0)), ((16 / 9, 39.6 * 9 / 16), (0.5, 0.3, -0.2))])
def test_hip_almeida_threshold_65536_vs_65537(ctx, cam_rot):
    """The same field one record either side of the switch from IEEE division to reciprocal multiplication
    (almeida.hip): both answers within 2e-6 of the oracle's and of each other."""
    (aspect, fov), eul = cam_rot
    n = 257
    xs, ys = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    pos = np.stack([(xs.ravel() + 0.5) / n, (ys.ravel() + 0.5) / n], 1)
    d = im.delta(pos, aspect, fov, im.euler_rot3(*np.radians(eul)))
    e = np.concatenate([pos, d], 1).astype(np.float32)
    cam = oracle.camera(aspect, fov)
    qs = []
    for m in (65536, 65537):
        q, _ = ctx.almeida(e[:m], aspect, fov, use_ransac=False)
        np.testing.assert_allclose(q, oracle.solve_ypr_given(e[:m], cam), atol=2e-6, rtol=0)
        qs.append(q)
    np.testing.assert_allclose(qs[0], qs[1], atol=2e-6, rtol=0)


# ---- BASELINE configs[2] at full size: 1080p dense flow, every pixel vs the oracle -----------------------------------
def test_hip_lk_flow_1080p_bit_exact(ctx):
    fr = synth.luma_sequence(2, 1920, 1080, max_step=6, seed=synth.SEED0 + 77)
    f_o = oracle.lk_flow(fr[0], fr[1], 3, 4, 3)
    f_g = ctx.lk_flow(fr[0], fr[1], 3, 4, 3)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


def test_hip_lk_flow_1080p_large_motion_and_flat_areas(ctx):
    """Tiles whose flows disagree by more than the staged rectangle allows (region jumps of +-16) and flat regions
    (singular structure tensors) take the non-staged path of the step kernel: same bits."""
    fr = synth.flatten_regions(synth.luma_sequence(2, 1920, 1080, max_step=16, seed=synth.SEED0 + 78), region=96)
    f_o = oracle.lk_flow(fr[0], fr[1], 3, 4, 3)
    f_g = ctx.lk_flow(fr[0], fr[1], 3, 4, 3)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


# ---- BASELINE configs[4] at full size: the fused per-frame path on 1080p frames ---------------------------------------
def test_hip_push_frame_1080p_matches_stagewise_oracle(ctx):
    W, H, F = 1920, 1080, 4
    fr = synth.luma_sequence(F, W, H, max_step=16, seed=synth.SEED0 + 79)
    cam = oracle.camera(16 / 9, 22.275)
    ctx.reset_frames()
    for k in range(F):
        r = ctx.push_frame(fr[k], block=16, search_range=16, aspect=16 / 9, fov_y_deg=22.275, want_entries=True, want_field=True,
                           use_ransac=bool(k & 1), num_iters=60, seed=100 + k)
        if k == 0:
            assert not r["have_vectors"]
            continue
        ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], 16, 16, threads=8)
        assert r["n_vectors"] == 8040
        np.testing.assert_array_equal(r["entries"].view(np.uint32), ent_o.view(np.uint32))
        det_o = oracle.detect_motion(ent_o)
        assert (r["motion"] is None) == (det_o is None)
        if det_o is not None:
            assert r["motion"][0] == det_o[0]
            np.testing.assert_array_equal(r["motion"][1].view(np.uint32), det_o[1].view(np.uint32))
        q_o = oracle.solve_ypr_ransac(ent_o, cam, 60, 0.05, 1000, seed=100 + k) if k & 1 else oracle.solve_ypr_given(ent_o, cam)
        np.testing.assert_allclose(r["quat"], q_o, atol=1e-4 if k & 1 else 2e-6, rtol=0)


# ---- cluster solver: what happens when its workgroups are not all there ------------------------------------------------
def test_hip_almeida_cluster_timeout_falls_back_to_the_stepped_solver(ctx, monkeypatch):
    """One workgroup withholds a granule (test hook): the others give up after their bounded spin, the kernel returns
    NaN, and the host-pointer entry point re-solves with one launch per step -- same answer, no hang."""
    import time
    e = synth.rotation_field(480, 270)                       # 129,600 records -> 127 workgroups
    cam = oracle.camera(16 / 9, 22.275)
    q_ok, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    monkeypatch.setenv("OFPS_HIP_ALMEIDA_TEST_FAULT", "2")
    t0 = time.perf_counter()
    q_fb, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    dt = time.perf_counter() - t0
    monkeypatch.delenv("OFPS_HIP_ALMEIDA_TEST_FAULT")
    assert np.isfinite(q_fb).all() and dt < 20.0
    np.testing.assert_allclose(q_fb, oracle.solve_ypr_given(e, cam), atol=2e-6, rtol=0)
    np.testing.assert_allclose(q_fb, q_ok, atol=2e-6, rtol=0)
    q_again, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)          # and the cluster path works again afterwards
    np.testing.assert_allclose(q_again, q_ok, atol=0, rtol=0)


@pytest.mark.parametrize("shape", [(120, 67), (480, 270), (960, 540)])
def test_hip_almeida_cluster_two_level_gather_matches_the_flat_gather(ctx, monkeypatch, shape):
    """Steps >= 1 of launches with >= 32 workgroups gather per XCD first (plain stores found in the XCD's L2), then across
    the XCDs; OFPS_HIP_ALMEIDA_HIER=0 forces the flat all-to-all gather, =2 the two-level one at any size.  Both are fixed
    summation orders of the same partials: each within 2e-6 of the oracle and of the other; a withheld granule times
    out in the two-level form as well."""
    e = synth.rotation_field(*shape)
    cam = oracle.camera(16 / 9, 22.275)
    q_o = oracle.solve_ypr_given(e, cam)
    qs = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("OFPS_HIP_ALMEIDA_HIER", mode)
        qs[mode], _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
        np.testing.assert_allclose(qs[mode], q_o, atol=2e-6, rtol=0)
    np.testing.assert_allclose(qs["0"], qs["2"], atol=2e-6, rtol=0)
    monkeypatch.setenv("OFPS_HIP_ALMEIDA_TEST_FAULT", "2")
    q_fb, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    monkeypatch.delenv("OFPS_HIP_ALMEIDA_TEST_FAULT")
    np.testing.assert_allclose(q_fb, q_o, atol=2e-6, rtol=0)


@pytest.mark.parametrize("fast", ["0", "1"])
def test_hip_almeida_arithmetic_switch_stays_inside_the_parity_bound(ctx, monkeypatch, fast):
    """OFPS_HIP_ALMEIDA_FAST forces the exact (IEEE division, unfused) or the folded arithmetic of the cluster solver at any
    field size (A/B runs): either way the quaternion stays within 2e-6 of the oracle's, below and above the 65,536 switch."""
    monkeypatch.setenv("OFPS_HIP_ALMEIDA_FAST", fast)
    cam = oracle.camera(16 / 9, 22.275)
    for shape in ((120, 67), (480, 270)):
        e = synth.rotation_field(*shape)
        q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
        np.testing.assert_allclose(q, oracle.solve_ypr_given(e, cam), atol=2e-6, rtol=0)


# ---- read-ahead form of the per-frame path: same bits as the synchronous call, two tickets in flight -------------------
def test_hip_push_frame_async_matches_sync_and_oracle(ctx):
    W, H, F = 1920, 1080, 6
    fr = synth.luma_sequence(F, W, H, max_step=16, seed=synth.SEED0 + 80)
    cam = oracle.camera(16 / 9, 22.275)
    kw = dict(block=16, search_range=16, aspect=16 / 9, fov_y_deg=22.275)
    # synchronous reference run
    ctx.reset_frames()
    sync = [ctx.push_frame(fr[k], want_entries=True, want_field=True, **kw) for k in range(F)]
    # asynchronous: frames staged in three page-locked buffers, results into page-locked arrays, tickets collected one late
    ctx.reset_frames()
    pins = [ctx.pinned_frame(H, W) for _ in range(3)]
    ents = [ctx.pinned_array((8040, 4)) for _ in range(2)]
    flds = [ctx.pinned_array((14, 14, 2)) for _ in range(2)]
    got, pending = [], None
    for k in range(F):
        np.copyto(pins[k % 3], fr[k])
        t = ctx.push_frame_async(pins[k % 3], out_entries=ents[k % 2], out_field=flds[k % 2], **kw)
        if pending is not None:
            r = ctx.frame_wait(pending[0]); r["entries"] = ents[pending[1] % 2].copy(); r["field"] = flds[pending[1] % 2].copy(); got.append(r)
        pending = (t, k)
    r = ctx.frame_wait(pending[0]); r["entries"] = ents[pending[1] % 2].copy(); r["field"] = flds[pending[1] % 2].copy(); got.append(r)
    with pytest.raises(Exception):
        ctx.frame_wait(pending[0])                                   # a ticket can be collected once
    assert not got[0]["have_vectors"] and not sync[0]["have_vectors"]
    for k in range(1, F):
        a, s = got[k], sync[k]
        assert a["have_vectors"] and a["n_vectors"] == 8040
        np.testing.assert_array_equal(a["entries"].view(np.uint32), s["entries"].view(np.uint32))
        np.testing.assert_array_equal(a["quat"].view(np.uint32), s["quat"].view(np.uint32))
        assert (a["motion"] is None) == (s["motion"] is None)
        if s["motion"] is not None:
            assert a["motion"][0] == s["motion"][0]
            np.testing.assert_array_equal(a["field"].view(np.uint32), s["motion"][1].view(np.uint32))
        ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], 16, 16, threads=8)
        np.testing.assert_array_equal(a["entries"].view(np.uint32), ent_o.view(np.uint32))
        np.testing.assert_allclose(a["quat"], oracle.solve_ypr_given(ent_o, cam), atol=2e-6, rtol=0)
    for b in pins + ents + flds:
        ctx.free_pinned(b)


def test_hip_push_frame_async_refuses_a_third_frame_in_flight(ctx):
    fr = synth.luma_sequence(3, 320, 192, max_step=8)
    ctx.reset_frames()
    pins = [ctx.pinned_frame(192, 320) for _ in range(3)]
    for k in range(3):
        np.copyto(pins[k], fr[k])
    t0 = ctx.push_frame_async(pins[0], search_range=8)
    t1 = ctx.push_frame_async(pins[1], search_range=8)
    with pytest.raises(Exception, match="collected"):
        ctx.push_frame_async(pins[2], search_range=8)
    assert not ctx.frame_wait(t0)["have_vectors"]
    assert ctx.frame_wait(t1)["have_vectors"]
    t2 = ctx.push_frame_async(pins[2], search_range=8)
    assert ctx.frame_wait(t2)["n_vectors"] == 20 * 12
    ctx.reset_frames()
    for b in pins:
        ctx.free_pinned(b)


# ---- add_vector_weighted through the C ABI (motion_field.rs:164-178) ---------------------------------------------------
def test_hip_densify_weighted_matches_oracle_and_hand_value(ctx):
    e = np.array([[0.5, 0.5, 0.25, -0.5], [0.5, 0.5, 1.0, 2.0]], np.float32)
    f = ctx.densify_weighted(e, [0.5, 0.25], 3, 3)
    # counts: eps + .5 + .25 = 0.75000012; sums: .25*.5 = .125, then 1*.25 + .125 = .375 -> .375 / .75000012
    assert f[1, 1, 0] == np.float32(0.375) / np.float32(0.75000012) and f[1, 1, 1] == np.float32(0.25) / np.float32(0.75000012)
    rng = np.random.default_rng(3)
    ent = np.concatenate([rng.uniform(0, 1, (5000, 2)), rng.normal(0, 0.01, (5000, 2))], 1).astype(np.float32)
    wgt = rng.uniform(0.0, 2.0, 5000).astype(np.float32)
    got, cells = ctx.densify_weighted(ent, wgt, 14, 14, want_cells=True)
    np.testing.assert_array_equal(got.view(np.uint32), oracle.densify_weighted(ent, wgt, 14, 14).view(np.uint32))
    # weights of exactly 1 are add_vector
    np.testing.assert_array_equal(ctx.densify_weighted(ent, np.ones(5000, np.float32), 14, 14).view(np.uint32),
                                  ctx.densify(ent, 14, 14).view(np.uint32))


def test_hip_ransac_batch_items_draw_different_samples(ctx):
    """Every item of a batched RANSAC call uses seed + item (the reference draws from thread_rng per call)."""
    import torch
    e = synth.rotation_field(120, 68, outlier_frac=0.3)
    d = torch.from_numpy(np.stack([e, e, e])).cuda()
    q = torch.empty((3, 4), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    try:
        ctx.almeida_dev(d.data_ptr(), e.shape[0], 3, 16 / 9, 22.275, True, 200, 0.05, 1000, 42, q.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    q = q.cpu().numpy()
    cam = oracle.camera(16 / 9, 22.275)
    for b in range(3):                                   # item b == a single call with seed 42 + b, which the oracle reproduces
        np.testing.assert_allclose(q[b], oracle.solve_ypr_ransac(e, cam, 200, 0.05, 1000, seed=42 + b), atol=1e-4, rtol=0)
    assert not (np.array_equal(q[0], q[1]) and np.array_equal(q[1], q[2]))


def test_hip_new_entry_points_reject_nonsense_loudly(ctx):
    from ofps_amd.runtime import OfpsHipError
    with pytest.raises(OfpsHipError, match="not in flight"):
        ctx.frame_wait(12345)
    e = np.zeros((4, 4), np.float32)
    with pytest.raises(AssertionError):
        ctx.densify_weighted(e, np.ones(3, np.float32), 4, 4)                       # one weight per entry
    with pytest.raises(OfpsHipError, match="bad grid"):
        ctx.densify_weighted(e, np.ones(4, np.float32), 0, 4)
    # a geometry change while a ticket is in flight drains the stream position instead of mixing frame sizes
    ctx.reset_frames()
    a = ctx.pinned_frame(192, 320); b = ctx.pinned_frame(96, 160)
    a[:] = 7; b[:] = 9
    t0 = ctx.push_frame_async(a, search_range=8)
    t1 = ctx.push_frame_async(b, search_range=8)                                     # restarts: first frame of a new stream
    r1 = ctx.frame_wait(t1)
    assert not r1["have_vectors"]
    ctx.reset_frames()
    ctx.free_pinned(a); ctx.free_pinned(b)
