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
    assert ctx.lk_helped_tiles() == 0          # every tile of every flow on this context found its parent tile's flows in time


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
# The reference's estimator cannot fail (singular LU -> zero step, almeida-estimator/src/lib.rs:181-185; < 3 inliers ->
# identity, :246-250).  The cluster kernel's bounded spin can expire (another process holding CUs); the LAST workgroup to
# leave then solves the item alone inside the same launch.  The fault injector (one workgroup withholds its step-3
# granule) exists only in libofps_hip_testhooks.so; the product library refuses to arm it.
@pytest.fixture(scope="module")
def hooks():
    from ofps_amd.runtime import HipContext
    c = HipContext(0, test_hooks=True)
    yield c
    c.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", None)
    c.close()


def test_product_library_refuses_the_fault_injectors(ctx):
    from ofps_amd._lib import OfpsHipError
    for name in ("OFPS_HIP_ALMEIDA_TEST_FAULT", "OFPS_HIP_LK_TEST_FALL", "OFPS_HIP_LK_TEST_WAIT_BUDGET"):
        with pytest.raises(OfpsHipError) as ei:
            ctx.set_option(name, "2")
        assert ei.value.code == -3                     # OFPS_HIP_EUNSUPPORTED
    with pytest.raises(OfpsHipError):
        ctx.set_option("OFPS_HIP_NO_SUCH_SWITCH", "1")


def test_hip_almeida_cluster_timeout_is_recovered_inside_the_launch(hooks):
    """Host-pointer entry: same answer as the undisturbed launch and the oracle, no NaN, no hang; the recovery counter
    says the in-kernel solve ran; the cluster path works again afterwards."""
    import time
    e = synth.rotation_field(480, 270)                       # 129,600 records -> 127 workgroups
    cam = oracle.camera(16 / 9, 22.275)
    q_ok, _ = hooks.almeida(e, 16 / 9, 22.275, use_ransac=False)
    r0 = hooks.almeida_recoveries()
    hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", "2")
    t0 = time.perf_counter()
    q_fb, _ = hooks.almeida(e, 16 / 9, 22.275, use_ransac=False)
    dt = time.perf_counter() - t0
    hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", None)
    assert hooks.almeida_recoveries() == r0 + 1
    assert np.isfinite(q_fb).all() and dt < 20.0
    np.testing.assert_allclose(q_fb, oracle.solve_ypr_given(e, cam), atol=2e-6, rtol=0)
    np.testing.assert_allclose(q_fb, q_ok, atol=2e-6, rtol=0)
    q_again, _ = hooks.almeida(e, 16 / 9, 22.275, use_ransac=False)          # and the cluster path works again afterwards
    np.testing.assert_allclose(q_again, q_ok, atol=0, rtol=0)
    assert hooks.almeida_recoveries() == r0 + 1


def test_hip_almeida_dev_batch_timeout_is_recovered_per_item(hooks):
    """Device-pointer entry (never synchronises, so nothing on the host could re-solve): batch of 3 fields, every item's
    cluster loses a granule, every item comes back finite and within 2e-6 of the oracle."""
    import torch
    shape = (160, 90)                                        # 14,400 records per item: cluster path for batches (> 8,192)
    fields = [synth.rotation_field(*shape, euler_deg=(0.5 + 0.2 * k, 0.3 - 0.1 * k, -0.2 + 0.15 * k), seed=synth.SEED0 + 40 + k)
              for k in range(3)]
    cam = oracle.camera(16 / 9, 22.275)
    d_ent = torch.from_numpy(np.stack(fields)).cuda()
    d_q = torch.full((3, 4), float("nan"), dtype=torch.float32, device="cuda")
    r0 = hooks.almeida_recoveries()
    hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", "1")
    hooks.use_torch_stream()
    try:
        hooks.almeida_dev(d_ent.data_ptr(), shape[0] * shape[1], 3, 16 / 9, 22.275, False, 200, 0.05, 1000, 0, d_q.data_ptr())
        torch.cuda.synchronize()
    finally:
        hooks.use_own_stream()
        hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", None)
    q = d_q.cpu().numpy()
    assert np.isfinite(q).all()
    assert hooks.almeida_recoveries() == r0 + 3
    for k in range(3):
        np.testing.assert_allclose(q[k], oracle.solve_ypr_given(fields[k], cam), atol=2e-6, rtol=0)


def test_hip_push_frame_async_ticket_survives_a_cluster_timeout(hooks):
    """cfg5's product path: a 1080p ticket (8,040 block vectors -> 8-workgroup cluster) whose estimator launch loses a
    granule still hands frame_wait the oracle's quaternion -- never NaN."""
    W, H = 1920, 1080
    fr = synth.luma_sequence(3, W, H, max_step=16, seed=synth.SEED0 + 81)
    cam = oracle.camera(16 / 9, 22.275)
    kw = dict(block=16, search_range=16, aspect=16 / 9, fov_y_deg=22.275)
    pinned = [hooks.pinned_array((H, W), np.uint8) for _ in range(3)]
    for k in range(3):
        pinned[k][...] = fr[k]
    hooks.reset_frames()
    r0 = hooks.almeida_recoveries()
    hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", "3")
    try:
        t0 = hooks.push_frame_async(pinned[0], **kw)
        res0 = hooks.frame_wait(t0)
        assert not res0["have_vectors"]
        t1 = hooks.push_frame_async(pinned[1], **kw)
        t2 = hooks.push_frame_async(pinned[2], **kw)
        res = [hooks.frame_wait(t1), hooks.frame_wait(t2)]
    finally:
        hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", None)
    assert hooks.almeida_recoveries() == r0 + 2
    for k, r in enumerate(res):
        assert r["have_vectors"] and np.isfinite(r["quat"]).all()
        ent_o, _ = oracle.sad_flow(fr[k], fr[k + 1], 16, 16)
        np.testing.assert_allclose(r["quat"], oracle.solve_ypr_given(ent_o, cam), atol=2e-6, rtol=0)
    # the synchronous form too
    hooks.reset_frames()
    hooks.push_frame(fr[0], **kw)
    hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", "1")
    try:
        r = hooks.push_frame(fr[1], **kw)
    finally:
        hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", None)
    ent_o, _ = oracle.sad_flow(fr[0], fr[1], 16, 16)
    np.testing.assert_allclose(r["quat"], oracle.solve_ypr_given(ent_o, cam), atol=2e-6, rtol=0)


@pytest.mark.parametrize("shape", [(120, 67), (480, 270), (960, 540)])
def test_hip_almeida_cluster_two_level_gather_matches_the_flat_gather(hooks, shape):
    """Steps >= 1 of launches with >= 32 workgroups gather per XCD first (plain stores found in the XCD's L2), then across
    the XCDs; OFPS_HIP_ALMEIDA_HIER=0 forces the flat all-to-all gather, =2 the two-level one at any size.  Both are fixed
    summation orders of the same partials: each within 2e-6 of the oracle and of the other; a withheld granule is
    recovered in the two-level form as well."""
    e = synth.rotation_field(*shape)
    cam = oracle.camera(16 / 9, 22.275)
    q_o = oracle.solve_ypr_given(e, cam)
    qs = {}
    try:
        for mode in ("0", "2"):
            hooks.set_option("OFPS_HIP_ALMEIDA_HIER", mode)
            qs[mode], _ = hooks.almeida(e, 16 / 9, 22.275, use_ransac=False)
            np.testing.assert_allclose(qs[mode], q_o, atol=2e-6, rtol=0)
        np.testing.assert_allclose(qs["0"], qs["2"], atol=2e-6, rtol=0)
        hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", "2")
        q_fb, _ = hooks.almeida(e, 16 / 9, 22.275, use_ransac=False)
    finally:
        hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", None)
        hooks.set_option("OFPS_HIP_ALMEIDA_HIER", None)
    np.testing.assert_allclose(q_fb, q_o, atol=2e-6, rtol=0)


@pytest.mark.parametrize("shape,batch", [((120, 67), 1), ((150, 84), 1), ((60, 40), 1), ((240, 135), 1), ((80, 50), 2), ((64, 60), 3)])
def test_hip_almeida_one_xcd_cluster_has_the_flat_exchange_s_bits(hooks, shape, batch):
    """Round 6: clusters of at most 64 256-thread workgroups are launched eight times as wide, every eighth workgroup works (one
    XCD under round-robin dispatch, verified in step 0 from the workgroups' XCC_IDs) and the steps' partial sums go through
    that XCD's L2.  OFPS_HIP_ALMEIDA_ONE_XCD=0 keeps the flat write-through exchange: the same numbers added in the same
    order -> the same bits, per item of a batch too; a withheld granule is recovered inside the launch in this form as well."""
    import torch
    n = shape[0] * shape[1]
    fields = [synth.rotation_field(*shape, euler_deg=(0.5 + 0.2 * k, 0.3 - 0.1 * k, -0.2 + 0.15 * k), seed=synth.SEED0 + 60 + k) for k in range(batch)]
    cam = oracle.camera(16 / 9, 22.275)
    d_ent = torch.from_numpy(np.stack(fields)).cuda()
    out = {}
    hooks.use_torch_stream()
    try:
        hooks.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster")
        for mode in ("0", "1"):
            hooks.set_option("OFPS_HIP_ALMEIDA_ONE_XCD", mode)
            d_q = torch.full((batch, 4), float("nan"), dtype=torch.float32, device="cuda")
            hooks.almeida_dev(d_ent.data_ptr(), n, batch, 16 / 9, 22.275, False, 0, 0.05, 0, 0, d_q.data_ptr())
            torch.cuda.synchronize()
            out[mode] = d_q.cpu().numpy()
        r0 = hooks.almeida_recoveries()
        hooks.set_option("OFPS_HIP_ALMEIDA_TEST_FAULT", "2")
        d_q = torch.full((batch, 4), float("nan"), dtype=torch.float32, device="cuda")
        hooks.almeida_dev(d_ent.data_ptr(), n, batch, 16 / 9, 22.275, False, 0, 0.05, 0, 0, d_q.data_ptr())
        torch.cuda.synchronize()
        q_fb = d_q.cpu().numpy()
        assert hooks.almeida_recoveries() == r0 + batch
    finally:
        hooks.use_own_stream()
        for k in ("OFPS_HIP_ALMEIDA_TEST_FAULT", "OFPS_HIP_ALMEIDA_ONE_XCD", "OFPS_HIP_ALMEIDA_PATH"): hooks.set_option(k, None)
    np.testing.assert_array_equal(out["0"].view(np.uint32), out["1"].view(np.uint32))
    for k in range(batch):
        q_o = oracle.solve_ypr_given(fields[k], cam)
        np.testing.assert_allclose(out["1"][k], q_o, atol=2e-6, rtol=0)
        np.testing.assert_allclose(q_fb[k], q_o, atol=2e-6, rtol=0)


@pytest.mark.parametrize("shape,block,ept", [((120, 67), 256, 1), ((120, 67), 256, 2), ((120, 67), 256, 4), ((120, 67), 1024, 1), ((120, 67), 1024, 4),
                                             ((60, 40), 256, 1), ((240, 135), 256, 2), ((240, 135), 256, 4), ((240, 135), 1024, 1),
                                             ((320, 180), 256, 4), ((320, 180), 1024, 2)])
def test_hip_almeida_cluster_every_workgroup_shape_inside_the_parity_bound(ctx, shape, block, ept):
    """Round 3's cluster launches pick 256-thread workgroups (1, 2 or 4 records per thread, at most 64 per item, flat gather
    with a cache line per granule) for block-vector sized fields and 1024-thread ones elsewhere; OFPS_HIP_ALMEIDA_BLOCK /
    _EPT force either.  Every shape is a fixed summation order of the same terms: within 2e-6 of the oracle, and of the
    shape the library picks by itself."""
    e = synth.rotation_field(*shape)
    q_o = oracle.solve_ypr_given(e, oracle.camera(16 / 9, 22.275))
    rec0 = ctx.almeida_recoveries()
    try:
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster")
        q_auto, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
        ctx.set_option("OFPS_HIP_ALMEIDA_BLOCK", block); ctx.set_option("OFPS_HIP_ALMEIDA_EPT", ept)
        q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    finally:
        for k in ("OFPS_HIP_ALMEIDA_PATH", "OFPS_HIP_ALMEIDA_BLOCK", "OFPS_HIP_ALMEIDA_EPT"): ctx.set_option(k, None)
    assert np.isfinite(q).all()
    np.testing.assert_allclose(q, q_o, atol=2e-6, rtol=0)
    np.testing.assert_allclose(q, q_auto, atol=2e-6, rtol=0)
    assert ctx.almeida_recoveries() == rec0                  # the workgroups were all there: nobody had to finish alone


def test_hip_almeida_cluster_launches_of_changing_shapes_back_to_back_are_reproducible(ctx):
    """The granule buffer is shared by all cluster launches of a context and never cleared between them: tags advance per
    launch, layouts differ with the workgroup count (packed granules up to 8 workgroups, a cache line each above, XCD slabs
    of the two-level gather).  2,300 launches of five shapes in rotation -- across the 16-bit tag wrap, where the buffer IS
    cleared -- must each return the bits their shape returned the first time, and nobody may have had to finish alone."""
    import torch
    shapes = [(120, 67), (60, 40), (80, 45), (240, 135), (480, 270)]        # 32, 10, 15 workgroups flat; 64 flat; 127 two-level
    fields = [torch.from_numpy(synth.rotation_field(w, h)).cuda() for (w, h) in shapes]
    outs = [torch.empty((1, 4), dtype=torch.float32, device="cuda") for _ in shapes]
    first = [None] * len(shapes)
    rec0 = ctx.almeida_recoveries()
    ctx.use_torch_stream()
    for k in range(2300):
        i = k % len(shapes)
        n = shapes[i][0] * shapes[i][1]
        ctx.almeida_dev(fields[i].data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, outs[i].data_ptr())
        if k < len(shapes) or k % 97 == 0 or k > 2290:
            q = outs[i].cpu().numpy().copy()
            assert np.isfinite(q).all()
            if first[i] is None: first[i] = q
            else: np.testing.assert_array_equal(q.view(np.uint32), first[i].view(np.uint32))
    torch.cuda.synchronize()
    assert ctx.almeida_recoveries() == rec0
    cam = oracle.camera(16 / 9, 22.275)
    for i, (w, h) in enumerate(shapes):
        np.testing.assert_allclose(first[i].ravel(), oracle.solve_ypr_given(synth.rotation_field(w, h), cam), atol=2e-6, rtol=0)


@pytest.mark.parametrize("fast", ["0", "1"])
def test_hip_almeida_arithmetic_switch_stays_inside_the_parity_bound(ctx, fast):
    """OFPS_HIP_ALMEIDA_FAST forces the exact (IEEE division, unfused) or the folded arithmetic of the cluster solver at any
    field size (A/B runs): either way the quaternion stays within 2e-6 of the oracle's, below and above the 65,536 switch."""
    ctx.set_option("OFPS_HIP_ALMEIDA_FAST", fast)
    try:
        cam = oracle.camera(16 / 9, 22.275)
        for shape in ((120, 67), (480, 270)):
            e = synth.rotation_field(*shape)
            q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
            np.testing.assert_allclose(q, oracle.solve_ypr_given(e, cam), atol=2e-6, rtol=0)
    finally:
        ctx.set_option("OFPS_HIP_ALMEIDA_FAST", None)


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


# ---- batched read-ahead form: n frames per ticket == n single pushes, bit for bit --------------------------------------
@pytest.mark.parametrize("use_ransac", [False, True])
def test_hip_push_frames_async_equals_single_pushes(ctx, use_ransac):
    """ofps_hip_push_frames_async: one upload, one search launch, one detector chain and one estimator launch per BATCH.
    Over a 1080p stream cut into batches of different sizes (one frame, many frames, a batch that makes the buffers grow)
    every frame's vectors and island equal what n calls of ofps_hip_push_frame return bit for bit, its quaternion to the
    solver's parity bound (RANSAC: frame j of a batch uses seed + j), and all of them the oracle's."""
    W, H, F = 1920, 1080, 12
    fr = synth.luma_sequence(F, W, H, max_step=16, seed=synth.SEED0 + 90)
    cam = oracle.camera(16 / 9, 22.275)
    kw = dict(block=16, search_range=16, aspect=16 / 9, fov_y_deg=22.275, use_ransac=use_ransac, num_iters=60)
    ctx.reset_frames()
    single = [ctx.push_frame(fr[k], seed=1000 + k, want_entries=True, **kw) for k in range(F)]
    ctx.reset_frames()
    nb = (W // 16) * (H // 16)
    cuts = [(0, 1), (1, 3), (3, 8), (8, 12)]                 # batch sizes 1, 2, 5 (buffers grow), 4
    pinned_frames = [ctx.pinned_array((b - a, H, W), np.uint8) for a, b in cuts]
    pinned_ent = [ctx.pinned_array((b - a, nb, 4), np.float32) for a, b in cuts]
    got = []
    pending = None
    for i, (a, b) in enumerate(cuts):
        pinned_frames[i][...] = fr[a:b]
        pinned_ent[i][...] = 0
        t = ctx.push_frames_async(pinned_frames[i], seed=1000 + a, out_entries=pinned_ent[i], **kw)
        if pending is not None:                              # two batches in flight
            got += [(r, e.copy()) for r, e in zip(ctx.frames_wait(pending[0]), pending[1])]
        pending = (t, pinned_ent[i])
    got += [(r, e.copy()) for r, e in zip(ctx.frames_wait(pending[0]), pending[1])]
    assert len(got) == F
    for k, ((r, ent), sref) in enumerate(zip(got, single)):
        assert r["have_vectors"] == sref["have_vectors"] == (k > 0)
        if k == 0:
            continue
        np.testing.assert_array_equal(ent.view(np.uint32), sref["entries"].view(np.uint32))
        assert (r["motion"] is None) == (sref["motion"] is None)
        if r["motion"] is not None:
            assert r["motion"][0] == sref["motion"][0]
        # the estimator of a batch is one launch over its items (one workgroup per item at this size), a lone frame's is the
        # 8-workgroup cluster solver: two fixed summation orders of the same terms, each within 2e-6 of the oracle
        np.testing.assert_allclose(r["quat"], sref["quat"], atol=1e-4 if use_ransac else 2e-6, rtol=0)
    # and the oracle, on two frames
    for k in (1, 9):
        ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], 16, 16)
        np.testing.assert_array_equal(got[k][1].view(np.uint32), ent_o.view(np.uint32))
        q_o = oracle.solve_ypr_ransac(ent_o, cam, 60, 0.05, 1000, seed=1000 + k) if use_ransac else oracle.solve_ypr_given(ent_o, cam)
        np.testing.assert_allclose(got[k][0]["quat"], q_o, atol=1e-4 if use_ransac else 2e-6, rtol=0)
    ctx.reset_frames()


def test_hip_interpolate_empty_cells_hand_derived_case(ctx):
    f = ctx.densify_interpolated(rv.INTERP_ENTRIES, 3, 1)
    assert [int(x) for x in np.asarray(f, np.float32).view(np.uint32).ravel()] == rv.INTERP_FIELD_BITS


def test_hip_ransac_at_the_image_centre_is_the_identity(ctx):
    """The hand-derived RANSAC case of tests/test_reference_vectors.py through the HIP path: every hypothesis and the refit
    are exactly singular (zero step, lib.rs:181-185), so the estimate is the identity quaternion, bit for bit; with fewer
    than 3 inliers it is the identity by lib.rs:246-250."""
    thr = np.float32(0.05) * (np.float32(np.pi) / np.float32(180.0))
    up = np.nextafter(thr, np.float32(1.0))
    motions = [(thr, 0), (0, thr), (up, 0), (0, 0), (-thr, 0), (thr, thr), (0, -up), (np.float32(0.5) * thr, np.float32(0.5) * thr)]
    e = np.array([[0.5, 0.5, mx, my] for mx, my in motions], np.float32)
    for sub in (slice(None), [2, 5, 6, 0]):
        q, _ = ctx.almeida(e[sub], 16 / 9, 22.275, use_ransac=True, num_iters=5, inlier_deg=0.05, num_samples=1000, seed=3)
        np.testing.assert_array_equal(q, np.array([1, 0, 0, 0], np.float32))


def test_hip_almeida_cluster_launches_of_two_contexts_are_chained_on_the_device(ctx):
    """A device with one live context launches its clusters ungated (no event behind the estimator's kernel); a second context switches the
    per-device gate on -- its ofps_hip_init drains the device, from then on every cluster launch waits for the other context's last one.
    Device-pointer calls of two contexts on two streams, never synchronised in between: every estimate is the oracle's, nobody had to finish
    alone (two clusters that each held part of the CUs would time out waiting for each other's workgroups), and the first context goes on
    when the second is gone."""
    import torch
    from ofps_amd.runtime import HipContext
    shapes = [(120, 67), (480, 270), (960, 540)]
    cam = oracle.camera(16 / 9, 22.275)
    fields = [synth.rotation_field(w, h, euler_deg=(0.4 + 0.1 * k, 0.2, -0.1 * k), seed=synth.SEED0 + 90 + k) for k, (w, h) in enumerate(shapes)]
    want = [oracle.solve_ypr_given(f, cam) for f in fields]
    d_fields = [torch.from_numpy(f).cuda() for f in fields]
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    rec0 = ctx.almeida_recoveries()
    ctx.set_stream(s_a.cuda_stream)
    other = HipContext(0)
    try:
        other.set_stream(s_b.cuda_stream)
        outs = []
        for rep in range(6):
            for k, (w, h) in enumerate(shapes):
                qa = torch.empty((1, 4), dtype=torch.float32, device="cuda"); qb = torch.empty((1, 4), dtype=torch.float32, device="cuda")
                ctx.almeida_dev(d_fields[k].data_ptr(), w * h, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, qa.data_ptr())
                other.almeida_dev(d_fields[(k + 1) % 3].data_ptr(), shapes[(k + 1) % 3][0] * shapes[(k + 1) % 3][1], 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, qb.data_ptr())
                outs.append((k, qa, (k + 1) % 3, qb))
        torch.cuda.synchronize()
        for k, qa, kb, qb in outs:
            np.testing.assert_allclose(qa.cpu().numpy().ravel(), want[k], atol=2e-6, rtol=0)
            np.testing.assert_allclose(qb.cpu().numpy().ravel(), want[kb], atol=2e-6, rtol=0)
        assert other.almeida_recoveries() == 0
    finally:
        other.close()
        ctx.use_own_stream()
    assert ctx.almeida_recoveries() == rec0
    q, _ = ctx.almeida(fields[0], 16 / 9, 22.275, use_ransac=False)
    np.testing.assert_allclose(q, want[0], atol=2e-6, rtol=0)
