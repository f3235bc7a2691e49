"""-m gpu property tests (hypothesis): the HIP densifier / detector against the CPU oracle on adversarial inputs --
positions outside [0,1], NaN / infinities, huge motions, degenerate grids."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from test_oracle_properties import entries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


@settings(max_examples=80, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(e=entries(max_n=200), w=st.integers(1, 40), h=st.integers(1, 40))
def test_hip_densify_matches_oracle_on_adversarial_inputs(ctx, e, w, h):
    f_g, cells_g = ctx.densify(e, w, h, want_cells=True)
    f_o, cells_o = oracle.densify(e, w, h, want_cells=True)
    np.testing.assert_array_equal(cells_g[: len(e)], cells_o[: len(e)])
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))        # NaN payloads included


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(e=entries(max_n=300), geom=st.sampled_from([(0.05, 3), (0.2, 2), (0.5, 1), (0.01, 4)]), target=st.sampled_from([0.0001, 0.003, 0.05]))
def test_hip_detect_matches_oracle_on_adversarial_inputs(ctx, e, geom, target):
    e = np.nan_to_num(e, nan=0.25, posinf=2.0, neginf=-1.0)
    e[:, 2:] = np.clip(e[:, 2:], -1e3, 1e3)
    a = ctx.detect(e, geom[0], geom[1], target)
    b = oracle.detect_motion(e, geom[0], geom[1], target)
    assert (a is None) == (b is None)
    if a is not None:
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(w=st.integers(1, 150), h=st.integers(1, 100), b=st.sampled_from([8, 16, 12]), r=st.sampled_from([0, 3, 4, 8, 12, 16, 20, 32]),
       seed=st.integers(0, 2**31 - 1), kind=st.sampled_from(["noise", "binary", "coarse", "flat"]))
def test_hip_sad_matches_oracle_on_random_geometry(ctx, w, h, b, r, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        fr = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    elif kind == "binary":                                   # saturated two-level content: exact ties everywhere
        fr = (rng.integers(0, 2, (2, h, w), dtype=np.uint8) * 255).astype(np.uint8)
    elif kind == "coarse":                                   # 4-level content shifted by a random vector
        base = (rng.integers(0, 4, (h + 64, w + 64), dtype=np.uint8) * 64).astype(np.uint8)
        dx, dy = int(rng.integers(-r, r + 1)), int(rng.integers(-r, r + 1))
        fr = np.stack([base[32:32 + h, 32:32 + w], base[32 + dy:32 + dy + h, 32 + dx:32 + dx + w]])
    else:
        fr = np.full((2, h, w), int(rng.integers(0, 256)), np.uint8)
    fr = np.ascontiguousarray(fr)
    ent_g, best_g = ctx.sad_flow(fr[0], fr[1], b, r, want_best=True)
    ent_o, best_o = oracle.sad_flow(fr[0], fr[1], b, r)
    np.testing.assert_array_equal(best_g, best_o)
    np.testing.assert_array_equal(ent_g.view(np.uint32), ent_o.view(np.uint32))


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(w=st.integers(2, 90), h=st.integers(2, 70), levels=st.integers(1, 4), radius=st.integers(1, 7), iters=st.integers(1, 3),
       seed=st.integers(0, 2**31 - 1), kind=st.sampled_from(["texture", "noise", "flat"]))
def test_hip_lk_flow_matches_oracle_on_random_geometry(ctx, w, h, levels, radius, iters, seed, kind):
    """Dense LK, every kernel variant (tiled radius 2/4/6 with the LDS-staged current frame, run-time radius otherwise),
    frames down to 2x2 with 4 pyramid levels, flows wild enough (noise) to force the global-memory fallback: bit-exact."""
    rng = np.random.default_rng(seed)
    if kind == "texture":
        fr = synth_pair(w, h, seed)
    elif kind == "noise":
        fr = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    else:
        fr = np.full((2, h, w), 99, np.uint8)
    f_g = ctx.lk_flow(fr[0], fr[1], levels, radius, iters)
    f_o = oracle.lk_flow(fr[0], fr[1], levels, radius, iters)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


def synth_pair(w, h, seed):
    from ofps_amd import synth
    return synth.luma_sequence(2, w, h, max_step=2, seed=seed % 100000)


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(nx=st.integers(2, 70), ny=st.integers(2, 50), rx=st.floats(-1.5, 1.5), ry=st.floats(-1.5, 1.5), rz=st.floats(-1.5, 1.5),
       fov=st.sampled_from([20.0, 39.6, 60.0, 90.0]), aspect=st.sampled_from([1.0, 4 / 3, 16 / 9]), seed=st.integers(0, 1000))
def test_hip_almeida_lsq_matches_oracle_on_random_rotations(ctx, nx, ny, rx, ry, rz, fov, aspect, seed):
    from ofps_amd import synth
    e = synth.rotation_field(nx, ny, euler_deg=(rx, ry, rz), aspect=aspect, fov_y_deg=fov, seed=seed)
    q_g, _ = ctx.almeida(e, aspect, fov, use_ransac=False)
    q_o = oracle.solve_ypr_given(e, oracle.camera(aspect, fov))
    np.testing.assert_allclose(q_g, q_o, atol=3e-6, rtol=0)


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(W=st.integers(1, 200), H=st.integers(1, 120), gw=st.integers(1, 160), gh=st.integers(1, 160), density=st.sampled_from([None, 0.0, 0.05, 0.5, 1.0]),
       seed=st.integers(0, 2**31 - 1), scale=st.sampled_from([1e-3, 1.0, 1e6]))
def test_hip_densify_raster_matches_oracle_on_random_geometry(ctx, W, H, gw, gh, density, seed, scale):
    """The rectangle walk of a per-pixel producer's records (ofps_hip_densify_raster_dev) == the oracle's add_vector loop on
    the (masked) records: frames from 1x1 to 200x120, grids coarser and finer than the frame, masks from empty to full,
    motions spanning nine orders of magnitude (the summation order shows in the low bits)."""
    import torch
    rng = np.random.default_rng(seed)
    flow = (rng.standard_normal((H, W, 2)) * scale).astype(np.float32)
    ent = oracle.flow_to_entries(flow)
    mask = None if density is None else (rng.random((H, W)) < density).astype(np.uint8)
    rec = ent if mask is None else ent[mask.reshape(-1) != 0]
    f_o = oracle.densify(rec, gw, gh) if len(rec) else np.zeros((gh, gw, 2), np.float32)
    d_ent = torch.from_numpy(ent).cuda()
    d_mask = None if mask is None else torch.from_numpy(mask).cuda()
    d_f = torch.full((gh, gw, 2), 3.0, dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    try:
        ctx.densify_raster_dev(d_ent.data_ptr(), None if mask is None else d_mask.data_ptr(), W, H, gw, gh, d_f.data_ptr(), verify=True)
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    np.testing.assert_array_equal(d_f.cpu().numpy().view(np.uint32), f_o.view(np.uint32))
