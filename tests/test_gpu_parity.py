"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bit-exact for integer/index work; floats to the tolerance written next to each check
(north_star: 1e-4 on float (u,v); most checks here are far tighter).
"""
import numpy as np
import pytest

from ofps_amd._lib import OfpsHipError

import oracle
from ofps_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def hooks_ctx():
    """libofps_hip_testhooks.so: the build whose fault injectors / forced rare paths can be armed."""
    from ofps_amd.runtime import HipContext
    c = HipContext(0, test_hooks=True)
    yield c
    c.close()


# ------------------------------------------------------------------ N1: SAD block matcher
SAD_CASES = [
    # (W, H, B, R, kind)
    (640, 360, 16, 8, "seq"),      # BASELINE configs[0] geometry
    (256, 144, 16, 16, "seq"),
    (320, 200, 8, 16, "seq"),
    (256, 136, 8, 32, "seq"),
    (256, 128, 16, 32, "seq"),
    (200, 120, 8, 8, "seq"),       # width not a multiple of 64 / of 4 blocks per workgroup
    (100, 70, 16, 16, "seq"),      # ragged: partial blocks dropped, W % 4 == 0 but W % 16 != 0
    (192, 96, 16, 16, "random"),   # unstructured noise: many near-ties
    (192, 96, 16, 16, "flat"),     # constant frames: every candidate ties at SAD 0 -> key order decides
    (256, 144, 16, 12, "seq"),     # ranges whose dx-group count is not a power of two: strip kernel with idle lanes
    (320, 208, 16, 20, "seq"),
    (256, 160, 16, 24, "seq"),
    (384, 192, 16, 28, "seq"),
    (200, 120, 8, 12, "seq"),
    (256, 136, 8, 20, "seq"),
    (256, 136, 8, 24, "random"),
    (320, 200, 8, 28, "seq"),
    (192, 96, 16, 24, "flat"),
    # two-level cells of 4x4 pixels: thousands of exact SAD ties between candidates with EQUAL d2 and different (dx, dy) --
    # the order inside a lane's key (rank code of the +-24..32 kernels, dy index of the narrower ones) decides the winner
    (256, 136, 8, 32, "blocky"),
    (320, 200, 8, 28, "blocky"),
    (256, 160, 16, 24, "blocky"),
    (256, 144, 16, 32, "blocky"),
    (256, 144, 16, 16, "blocky"),
    (192, 112, 16, 10, "seq"),     # range not a multiple of 4: generic kernel
    (96, 96, 12, 5, "seq"),        # generic kernel (block/range outside the packed-SAD table)
    (64, 48, 32, 4, "seq"),        # generic kernel, SAD beyond 16 bits possible
]


def _frames(W, H, R, kind):
    if kind == "seq":
        return synth.luma_sequence(2, W, H, max_step=R, seed=synth.SEED0 + W + H)
    if kind == "random":
        return synth.random_luma(2, W, H, seed=7)
    if kind == "blocky":
        rng = np.random.default_rng(W * 131 + H)
        cells = (rng.integers(0, 2, (2, (H + 3) // 4, (W + 3) // 4)) * 200 + 20).astype(np.uint8)
        return np.ascontiguousarray(np.repeat(np.repeat(cells, 4, axis=1), 4, axis=2)[:, :H, :W])
    return np.full((2, H, W), 77, np.uint8)


@pytest.mark.parametrize("W,H,B,R,kind", SAD_CASES)
def test_sad_matches_oracle_bit_exact(ctx, W, H, B, R, kind):
    fr = _frames(W, H, R, kind)
    ent_o, best_o = oracle.sad_flow(fr[0], fr[1], B, R)
    ent_g, best_g = ctx.sad_flow(fr[0], fr[1], B, R, want_best=True)
    assert best_g.shape == best_o.shape
    np.testing.assert_array_equal(best_g, best_o)           # (dx, dy, SAD): integers, exact
    np.testing.assert_array_equal(ent_g.view(np.uint32), ent_o.view(np.uint32))   # f32 records: same bits


PRUNED_CASES = [(256, 144, "seq"), (100, 70, "seq"), (192, 96, "random"), (192, 96, "flat"), (640, 368, "seq"), (1280, 720, "pan"),
                (272, 160, "gradient"), (1920, 1080, "seq")]


@pytest.mark.parametrize("W,H,kind", PRUNED_CASES)
def test_sad_pruned_mode_is_bit_identical_to_exhaustive_spec(ctx, W, H, kind):
    """Partial-distortion-elimination mode (include/ofps_hip.h OFPS_HIP_SAD_PRUNED) must return the exhaustive winner on
    any content: structured frames (few survivors), noise (survivor lists overflow -> strips fall back to the
    exhaustive kernel), constant frames (every bound ties at 0), smooth gradients (weak bounds), a camera pan."""
    if kind == "gradient":
        yy, xx = np.mgrid[0:H, 0:W]
        a = ((xx * 3 + yy * 2) % 256).astype(np.uint8)
        fr = np.stack([a, np.roll(a, (3, -5), (0, 1))])
    elif kind == "pan":                               # one global translation + sensor noise: almost nothing overflows
        fr = synth.luma_sequence(2, W, H, max_step=12, seed=3, region=4096, noise=1)
    else:
        fr = _frames(W, H, 16, kind)
    ctx.set_sad_mode(ctx.SAD_PRUNED)
    try:
        ent_g, best_g = ctx.sad_flow(fr[0], fr[1], 16, 16, want_best=True)
        overflow = ctx.sad_pruned_overflow_strips()
    finally:
        ctx.set_sad_mode(ctx.SAD_EXHAUSTIVE)
    if kind == "pan":
        assert overflow < 0.2 * ((W // 16 + 7) // 8) * (H // 16)          # the pruned kernel did the work, not the fallback
    if kind in ("random", "flat"):
        assert overflow > 0                                                # nothing to prune: the fallback is exercised
    ent_o, best_o = oracle.sad_flow(fr[0], fr[1], 16, 16, threads=8)
    np.testing.assert_array_equal(best_g, best_o)
    np.testing.assert_array_equal(ent_g.view(np.uint32), ent_o.view(np.uint32))


def test_sad_planted_displacement_recovered(ctx):
    """Size-independent property at full 1080p: a frame shifted by an integer vector must come back
    as that vector for every interior block (SAD 0 there), bit-exact positions."""
    W, H, B, R = 1920, 1080, 16, 16
    base = synth.luma_sequence(1, W + 64, H + 64, max_step=0, noise=0, seed=99)[0]
    dx, dy = 7, -11
    prev = np.ascontiguousarray(base[32:32 + H, 32:32 + W])
    cur = np.ascontiguousarray(base[32 + dy:32 + dy + H, 32 + dx:32 + dx + W])   # cur(x,y) = prev(x+dx, y+dy)
    ent, best = ctx.sad_flow(prev, cur, B, R, want_best=True)
    nbx, nby = W // B, H // B
    best = best.reshape(nby, nbx, 3)
    inner = best[1:-1, 1:-1]
    assert (inner[..., 0] == dx).all() and (inner[..., 1] == dy).all() and (inner[..., 2] == 0).all()
    ent = ent.reshape(nby, nbx, 4)
    np.testing.assert_array_equal(ent[1:-1, 1:-1, 2], np.float32(dx) * -np.float32(np.float32(1.0) / np.float32(W)))


def test_sad_identical_frames_zero_motion(ctx):
    fr = synth.luma_sequence(1, 640, 360, max_step=0)[0]
    _, best = ctx.sad_flow(fr, fr, 16, 16, want_best=True)
    assert (best == 0).all()


def test_sad_batched_device_path_matches_pairwise(ctx):
    """Batched device entry point (what bench.py times) == per-pair host entry point."""
    import torch
    W, H, B, R, F = 320, 192, 16, 16, 4
    fr = synth.luma_sequence(F, W, H, max_step=R)
    d = torch.from_numpy(fr).cuda()
    nb = (W // B) * (H // B)
    ctx.use_torch_stream()
    for ref_mode in (0, 1):
        out = torch.zeros((F - 1, nb, 4), dtype=torch.float32, device="cuda")
        best = torch.zeros((F - 1, nb, 3), dtype=torch.int32, device="cuda")
        ctx.sad_flow_dev(d.data_ptr(), F, W, H, W, W * H, ref_mode, B, R, out.data_ptr(), best.data_ptr())
        torch.cuda.synchronize()
        for k in range(F - 1):
            prev = fr[0] if ref_mode else fr[k]
            _, bo = oracle.sad_flow(prev, fr[k + 1], B, R)
            np.testing.assert_array_equal(best[k].cpu().numpy(), bo)
    ctx.use_own_stream()


def test_sad_cfg4_full_size_crop_consistency(ctx):
    """BASELINE configs[3] geometry at full size (3840x2160, 8x8 blocks, +-32): an exhaustive search is local, so every
    block whose +-32 window lies inside an aligned crop must get the same (dx, dy, SAD) from the oracle run on the crop
    alone -- a size-independent check the oracle finishes in seconds."""
    W, H, B, R = 3840, 2160, 8, 32
    fr = synth.luma_sequence(2, W, H, max_step=R, seed=41)
    _, best = ctx.sad_flow(fr[0], fr[1], B, R, want_best=True)
    best = best.reshape(H // B, W // B, 3)
    for cx, cy, cw, ch in [(0, 0, 320, 192), (1760, 984, 320, 192), (W - 320, H - 192, 320, 192), (2048, 0, 256, 160)]:
        _, bo = oracle.sad_flow(fr[0][cy:cy + ch, cx:cx + cw], fr[1][cy:cy + ch, cx:cx + cw], B, R, threads=8)
        bo = bo.reshape(ch // B, cw // B, 3)
        # blocks of the crop whose window is not cut by a crop edge that is not also a frame edge
        x_lo = 0 if cx == 0 else R // B; x_hi = cw // B if cx + cw == W else cw // B - R // B
        y_lo = 0 if cy == 0 else R // B; y_hi = ch // B if cy + ch == H else ch // B - R // B
        got = best[cy // B + y_lo:cy // B + y_hi, cx // B + x_lo:cx // B + x_hi]
        np.testing.assert_array_equal(got, bo[y_lo:y_hi, x_lo:x_hi])


def test_sad_bench_shape_batch_full_size(ctx):
    """The launch bench.py times (1080p, 16x16, +-16, consecutive pairs of one resident sequence, padded stride): every
    pair of the batch equals the per-pair host call, and sampled pairs equal the oracle."""
    import torch
    W, H, B, R, F = 1920, 1080, 16, 16, 9
    stride = 1984                                        # not the frame width: rows are padded
    fr = synth.luma_sequence(F, W, H, max_step=R, seed=synth.SEED0, stride=stride)
    d = torch.from_numpy(fr).cuda()
    nb = (W // B) * (H // B)
    out = torch.zeros((F - 1, nb, 4), dtype=torch.float32, device="cuda")
    best = torch.zeros((F - 1, nb, 3), dtype=torch.int32, device="cuda")
    ctx.use_torch_stream()
    try:
        ctx.sad_flow_dev(d.data_ptr(), F, W, H, stride, stride * H, 0, B, R, out.data_ptr(), best.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    out, best = out.cpu().numpy(), best.cpu().numpy()
    for k in range(F - 1):
        ent_h, best_h = ctx.sad_flow(fr[k][:, :W], fr[k + 1][:, :W], B, R, want_best=True)
        np.testing.assert_array_equal(best[k], best_h)
        np.testing.assert_array_equal(out[k].view(np.uint32), ent_h.view(np.uint32))
    for k in (0, 4, 7):
        ent_o, best_o = oracle.sad_flow(np.ascontiguousarray(fr[k][:, :W]), np.ascontiguousarray(fr[k + 1][:, :W]), B, R, threads=8)
        np.testing.assert_array_equal(best[k], best_o)
        np.testing.assert_array_equal(out[k].view(np.uint32), ent_o.view(np.uint32))


@pytest.mark.parametrize("B,R", [(16, 16), (16, 8), (16, 32), (8, 32), (8, 16), (8, 8), (16, 24)])
def test_sad_device_path_with_rows_not_16_byte_aligned(ctx, B, R):
    """Device-resident frames whose row stride is only 4-byte aligned cannot use the 16-byte staging of the strip kernel:
    the per-block kernel (table geometries) or the generic kernel (others) must return the same bits."""
    import torch
    W, H, stride = 324, 200, 332
    fr = synth.luma_sequence(3, W, H, max_step=min(R, 16), seed=B * 100 + R, stride=stride)
    d = torch.from_numpy(fr).cuda()
    nb = (W // B) * (H // B)
    best = torch.zeros((2, nb, 3), dtype=torch.int32, device="cuda")
    out = torch.zeros((2, nb, 4), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    try:
        ctx.sad_flow_dev(d.data_ptr(), 3, W, H, stride, stride * H, 0, B, R, out.data_ptr(), best.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    for k in range(2):
        ent_o, best_o = oracle.sad_flow(np.ascontiguousarray(fr[k][:, :W]), np.ascontiguousarray(fr[k + 1][:, :W]), B, R)
        np.testing.assert_array_equal(best[k].cpu().numpy(), best_o)
        np.testing.assert_array_equal(out[k].cpu().numpy().view(np.uint32), ent_o.view(np.uint32))


def test_sad_odd_frame_sizes_full_oracle(ctx):
    """Frame sizes that are not multiples of anything (1917 x 1079): the host entry point repacks rows, partial blocks are
    dropped, the right/bottom search windows are clipped by the true frame edge."""
    W, H = 1917, 1079
    fr = synth.luma_sequence(2, W, H, max_step=16, seed=5)
    for B, R in ((16, 16), (8, 16)):
        ent_o, best_o = oracle.sad_flow(fr[0], fr[1], B, R, threads=8)
        ent_g, best_g = ctx.sad_flow(fr[0], fr[1], B, R, want_best=True)
        np.testing.assert_array_equal(best_g, best_o)
        np.testing.assert_array_equal(ent_g.view(np.uint32), ent_o.view(np.uint32))


def test_sad_8k_frame_crop_consistency(ctx):
    """7680 x 4320 (the largest frame a 16-bit grid dimension still allows with room to spare): crop consistency against
    the oracle at the four corners and the centre."""
    W, H, B, R = 7680, 4320, 16, 16
    fr = synth.luma_sequence(2, W, H, max_step=R, seed=8, region=128)
    _, best = ctx.sad_flow(fr[0], fr[1], B, R, want_best=True)
    best = best.reshape(H // B, W // B, 3)
    cw, ch = 256, 160
    for cx, cy in [(0, 0), (W - cw, 0), (0, H - ch), (W - cw, H - ch), (3712, 2080)]:
        _, bo = oracle.sad_flow(fr[0][cy:cy + ch, cx:cx + cw], fr[1][cy:cy + ch, cx:cx + cw], B, R, threads=8)
        bo = bo.reshape(ch // B, cw // B, 3)
        x_lo = 0 if cx == 0 else R // B; x_hi = cw // B if cx + cw == W else cw // B - R // B
        y_lo = 0 if cy == 0 else R // B; y_hi = ch // B if cy + ch == H else ch // B - R // B
        np.testing.assert_array_equal(best[cy // B + y_lo:cy // B + y_hi, cx // B + x_lo:cx // B + x_hi], bo[y_lo:y_hi, x_lo:x_hi])


def test_sad_rejects_bad_arguments(ctx):
    from ofps_amd.runtime import OfpsHipError
    fr = np.zeros((32, 32), np.uint8)
    with pytest.raises(OfpsHipError):
        ctx.sad_flow(fr, fr, 0, 8)
    with pytest.raises(OfpsHipError):
        ctx.sad_flow(fr, fr, 16, 100)


# ------------------------------------------------------------------ A1-A4: densifier
def _entries(n, seed, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    e = np.empty((n, 4), np.float32)
    e[:, :2] = rng.uniform(lo, hi, (n, 2)).astype(np.float32)
    e[:, 2:] = rng.normal(0, 0.01, (n, 2)).astype(np.float32)
    return e


@pytest.mark.parametrize("n,w,h", [(1000, 14, 14), (1000, 150, 84), (8040, 14, 14), (50000, 160, 160),
                                   (1, 3, 3), (0, 4, 4), (5000, 1, 1), (3000, 256, 1), (70000, 16, 16)])
def test_densify_bit_exact(ctx, n, w, h):
    e = _entries(n, 100 + n + w)
    f_o, c_o = oracle.densify(e, w, h, want_cells=True)
    f_g, c_g = ctx.densify(e, w, h, want_cells=True)
    np.testing.assert_array_equal(c_g, c_o)                               # cell indices: exact
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))   # input-order sums: same bits


@pytest.mark.parametrize("n,w,h", [(1, 1, 1), (63, 14, 14), (64, 14, 14), (65, 16, 16), (880, 14, 14), (8040, 14, 14), (8192, 16, 16),
                                   (8192, 256, 1), (5000, 1, 1), (4097, 3, 85), (8193, 14, 14)])
@pytest.mark.parametrize("spread", ["uniform", "one_cell", "out_of_range"])
def test_densify_one_workgroup_path_matches_the_general_path_and_the_oracle(ctx, n, w, h, spread):
    """Up to 8,192 entries on up to 256 cells the densifier is one workgroup per item (densify_small_kernel); above that,
    or with OFPS_HIP_DENSIFY_NO_SMALL, the six-kernel stable sort.  Same bits either way, and the oracle's: slot edges
    (63 / 64 / 65 entries), the size limits on both sides, every entry in one cell, positions the clamp collapses."""
    e = _entries(n, 7 * n + w)
    if spread == "one_cell":
        e[:, :2] = (0.37, 0.61)
    elif spread == "out_of_range":
        e[::3, 0] = -0.2; e[1::5, 1] = 1.4; e[2::7, 0] = np.nan
    f_o = oracle.densify(e, w, h)
    f_small = ctx.densify(e, w, h)
    np.testing.assert_array_equal(f_small.view(np.uint32), f_o.view(np.uint32))
    ctx.set_option("OFPS_HIP_DENSIFY_NO_SMALL", "1")
    try:
        f_gen = ctx.densify(e, w, h)
    finally:
        ctx.set_option("OFPS_HIP_DENSIFY_NO_SMALL", None)
    np.testing.assert_array_equal(f_gen.view(np.uint32), f_o.view(np.uint32))


def test_densify_out_of_range_and_nan_positions(ctx):
    """nalgebra clamp quirk (semantics unverified against the Rust build): any coordinate
    <= 0 collapses the point to (0,0), any >= 1 to (1,1); NaN -> (0,0)."""
    e = _entries(64, 5, -0.5, 1.5)
    e[0, :2] = (np.nan, 0.5); e[1, :2] = (0.5, np.nan); e[2, :2] = (0.0, 0.5); e[3, :2] = (1.0, 0.2)
    e[4, :2] = (np.inf, 0.5); e[5, :2] = (-np.inf, 0.5); e[6, :2] = (0.999999, 0.999999)
    f_o, c_o = oracle.densify(e, 14, 9, want_cells=True)
    f_g, c_g = ctx.densify(e, 14, 9, want_cells=True)
    np.testing.assert_array_equal(c_g, c_o)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


def test_densify_per_pixel_entries(ctx):
    """cv-decoder full-res shape (scaled): every pixel contributes, 150x84 grid."""
    e = synth.rotation_field(480, 270)
    f_o = oracle.densify(e, 150, 84)
    f_g = ctx.densify(e, 150, 84)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    out_o = oracle.densify_to_entries(e, 150, 84)
    out_g = ctx.densify_to_entries(e, 150, 84)
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))


@pytest.mark.parametrize("n,w,h,seed", [(40, 12, 9, 8), (5, 16, 16, 1), (300, 40, 23, 2), (1, 7, 5, 3), (2000, 150, 84, 4)])
def test_interpolate_empty_cells_bit_exact(ctx, n, w, h, seed):
    """MotionFieldDensifier::interpolate_empty_cells (motion_field.rs:193-294) through the device path."""
    e = _entries(n, seed)
    np.testing.assert_array_equal(ctx.densify_interpolated(e, w, h).view(np.uint32),
                                  oracle.densify_interpolated(e, w, h).view(np.uint32))


def test_interpolate_empty_cells_no_vectors(ctx):
    assert (ctx.densify_interpolated(np.zeros((0, 4), np.float32), 6, 4) == 0).all()


# ------------------------------------------------------------------ A5: detector
def _island_entries(dim, cells_on, mag=0.01, per_cell=3, seed=0):
    """Entries landing exactly in chosen cells of a dim x dim grid."""
    rng = np.random.default_rng(seed)
    out = []
    for (x, y) in cells_on:
        for _ in range(per_cell):
            px = (x + rng.uniform(-0.2, 0.2)) / (dim - 1); py = (y + rng.uniform(-0.2, 0.2)) / (dim - 1)
            out.append((min(max(px, 1e-4), 1 - 1e-4), min(max(py, 1e-4), 1 - 1e-4), mag, 0.0))
    # background: one still vector in every cell
    for y in range(dim):
        for x in range(dim):
            out.append((min(max(x / (dim - 1), 1e-4), 1 - 1e-4), min(max(y / (dim - 1), 1e-4), 1 - 1e-4), 0.0, 0.0))
    return np.array(out, np.float32)


def _check_detect(ctx, e, **kw):
    r_o = oracle.detect_motion(e, **kw)
    r_g = ctx.detect(e, **kw)
    if r_o is None:
        assert r_g is None
        return None
    assert r_g is not None
    assert r_g[0] == r_o[0]                                               # area: exact
    np.testing.assert_array_equal((r_g[1] != 0).any(-1), (r_o[1] != 0).any(-1))   # membership: exact
    np.testing.assert_array_equal(r_g[1].view(np.uint32), r_o[1].view(np.uint32))
    return r_g


def test_detect_default_params_random_field(ctx):
    for seed in range(6):
        e = _entries(8040, 300 + seed)
        e[:, 2:] *= np.float32(30.0 if seed % 2 else 0.3)
        _check_detect(ctx, e)


def test_detect_seed_cell_omitted_and_tie_break(ctx):
    dim = oracle.block_dim(0.05, 3)
    assert dim == 14 == ctx.block_dim(0.05, 3)
    # two islands of equal area 12: the first in raster order must win; its seed cell stays zero
    a = [(x, 2) for x in range(1, 7)] + [(x, 3) for x in range(1, 7)]
    b = [(x, 9) for x in range(5, 11)] + [(x, 10) for x in range(5, 11)]
    e = _island_entries(dim, a + b, mag=0.05)
    r = _check_detect(ctx, e)
    assert r is not None and r[0] == 12
    assert (r[1][2, 1] == 0).all() and (r[1][2, 2] != 0).any()           # seed (1,2) omitted
    assert (r[1][9:11] == 0).all()                                        # second island not returned


def test_detect_min_size_boundary(ctx):
    dim = 14
    # area 9 of 196 = 0.0459 < 0.05 -> None ; area 10 = 0.051 -> Some
    nine = [(x, y) for x in range(3) for y in range(3)]
    assert _check_detect(ctx, _island_entries(dim, nine, mag=0.05)) is None
    ten = nine + [(3, 0)]
    r = _check_detect(ctx, _island_entries(dim, ten, mag=0.05))
    assert r is not None and r[0] == 10


def test_detect_diagonal_connectivity_and_large_grid(ctx):
    # min_size 0.01, subdivide 16 -> dim 160 (largest grid the properties allow)
    dim = oracle.block_dim(0.01, 16)
    assert dim == 160
    diag = [(i, i) for i in range(10, 90)] + [(i + 1, i) for i in range(10, 90, 7)]   # stays clear of the snake
    snake = [(x, 155) for x in range(0, 160)] + [(159, y) for y in range(100, 155)] + [(x, 100) for x in range(20, 160)]
    e = _island_entries(dim, diag + snake, mag=0.05, per_cell=1)
    r = _check_detect(ctx, e, min_size=0.01, subdivide=16, target_motion=0.003)
    assert r is not None and r[0] == len(set(snake))


def test_detect_empty_and_still_inputs(ctx):
    assert ctx.detect(np.zeros((0, 4), np.float32)) is None
    e = _entries(500, 1); e[:, 2:] = 0
    assert _check_detect(ctx, e) is None


# ------------------------------------------------------------------ A6-A12: Almeida
def test_almeida_reference_known_answer_lsq(ctx):
    """The reference's own test (almeida-estimator/src/lib.rs:359-364) through the HIP path, plus
    agreement with the oracle's quaternion (tolerance 1e-5 per component; north_star allows 1e-4)."""
    import almeida_cases as ac
    cam = oracle.camera(1.0, 90.0)
    for rot, ang, q, field in ac.cases():
        est, tr = ctx.almeida(field, 1.0, 90.0, use_ransac=False)
        err = ac.error_deg(q, est)
        assert err < 0.1 * rot or (rot == 0 and err == 0), (rot, ang, err)
        q_o = oracle.solve_ypr_given(field, cam)
        np.testing.assert_allclose(est, q_o, atol=1e-5, rtol=0)
        assert (tr == 0).all()


def test_almeida_reference_known_answer_ransac(ctx):
    """lib.rs:366-372 (100 iterations) through the HIP path; same seeds as the oracle."""
    import almeida_cases as ac
    cam = oracle.camera(1.0, 90.0)
    for i, (rot, ang, q, field) in enumerate(ac.cases()):
        est, _ = ctx.almeida(field, 1.0, 90.0, use_ransac=True, num_iters=100, inlier_deg=0.05, num_samples=1000,
                             seed=1234 + i)
        err = ac.error_deg(q, est)
        assert err < 0.1 * rot or (rot == 0 and err == 0), (rot, ang, err)
        q_o = oracle.solve_ypr_ransac(field, cam, 100, 0.05, 1000, seed=1234 + i)
        np.testing.assert_allclose(est, q_o, atol=1e-4, rtol=0)


@pytest.mark.parametrize("n_side", [(30, 20), (40, 40), (64, 36), (120, 67), (160, 90), (480, 270), (960, 540)])   # wg<1,2,4,8>, step, dense
def test_almeida_noisy_field_matches_oracle(ctx, n_side):
    """cfg3-shaped input (per-pixel entries, planted rotation + noise); covers the one-workgroup and the
    multi-launch solver.  Tolerance 2e-6 on quaternion components (sum order differs)."""
    w, h = n_side
    e = synth.rotation_field(w, h)
    cam = oracle.camera(16 / 9, 39.6 * 9 / 16)
    q_o = oracle.solve_ypr_given(e, cam)
    q_g, _ = ctx.almeida(e, 16 / 9, 39.6 * 9 / 16, use_ransac=False)
    np.testing.assert_allclose(q_g, q_o, atol=2e-6, rtol=0)


def test_almeida_ransac_with_outliers(ctx):
    e = synth.rotation_field(120, 68, outlier_frac=0.3)
    cam = oracle.camera(16 / 9, 39.6 * 9 / 16)
    q_o, inl = oracle.solve_ypr_ransac(e, cam, 200, 0.05, 1000, seed=42, want_inliers=True)
    q_g, _ = ctx.almeida(e, 16 / 9, 39.6 * 9 / 16, use_ransac=True, num_iters=200, inlier_deg=0.05, num_samples=1000, seed=42)
    assert len(inl) >= 3
    np.testing.assert_allclose(q_g, q_o, atol=1e-4, rtol=0)
    # and the answer is the planted rotation, not the outliers
    q_lsq_clean = oracle.solve_ypr_given(synth.rotation_field(120, 68), cam)
    assert np.degrees(oracle.quat_angle_to(q_lsq_clean, q_g)) < 0.02


def test_almeida_degenerate_inputs(ctx):
    ident = np.array([1, 0, 0, 0], np.float32)
    for n in (0, 1, 2):
        e = _entries(n, 9)
        for ransac in (False, True):
            q, _ = ctx.almeida(e, 1.0, 90.0, use_ransac=ransac, num_iters=10, num_samples=100, seed=1)
            cam = oracle.camera(1.0, 90.0)
            q_o = oracle.solve_ypr_ransac(e, cam, 10, 0.05, 100, seed=1) if ransac else oracle.solve_ypr_given(e, cam)
            np.testing.assert_allclose(np.abs(q), np.abs(q_o), atol=1e-5)
            if n == 0:
                np.testing.assert_allclose(np.abs(q), ident, atol=0)


def test_almeida_batched_device_path(ctx):
    import torch
    e = np.stack([synth.rotation_field(64, 36, euler_deg=(0.1 * k, -0.2, 0.05 * k), seed=k) for k in range(5)])
    d = torch.from_numpy(e).cuda()
    out = torch.zeros((5, 4), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    ctx.almeida_dev(d.data_ptr(), e.shape[1], 5, 16 / 9, 22.275, False, 0, 0.05, 0, 0, out.data_ptr())
    torch.cuda.synchronize()
    ctx.use_own_stream()
    cam = oracle.camera(16 / 9, 22.275)
    for k in range(5):
        np.testing.assert_allclose(out[k].cpu().numpy(), oracle.solve_ypr_given(e[k], cam), atol=2e-6, rtol=0)


# ------------------------------------------------------------------ fused per-frame path (cfg5)
def test_push_frame_matches_stagewise_oracle(ctx):
    """ofps_hip_push_frame == decoder -> detector + estimator run stage by stage on the oracle."""
    W, H, F = 640, 360, 5
    fr = synth.luma_sequence(F, W, H, max_step=8)
    cam = oracle.camera(16 / 9, 22.275)
    ctx.reset_frames()
    for k in range(F):
        r = ctx.push_frame(fr[k], block=16, search_range=8, aspect=16 / 9, fov_y_deg=22.275, want_entries=True, want_field=True)
        if k == 0:
            assert not r["have_vectors"] and r["motion"] is None and (r["quat"] == [1, 0, 0, 0]).all()
            continue
        ent_o, _ = oracle.sad_flow(fr[k - 1], fr[k], 16, 8)
        assert r["have_vectors"] and r["n_vectors"] == len(ent_o)
        np.testing.assert_array_equal(r["entries"].view(np.uint32), ent_o.view(np.uint32))
        det_o = oracle.detect_motion(ent_o)
        assert (r["motion"] is None) == (det_o is None)
        if det_o is not None:
            assert r["motion"][0] == det_o[0]
            np.testing.assert_array_equal(r["motion"][1].view(np.uint32), det_o[1].view(np.uint32))
        np.testing.assert_allclose(r["quat"], oracle.solve_ypr_given(ent_o, cam), atol=2e-6, rtol=0)
    # a geometry change restarts the stream
    r = ctx.push_frame(fr[0][:180, :320].copy(), block=16, search_range=8)
    assert not r["have_vectors"]


# ------------------------------------------------------------------ N2: dense pyramidal LK flow
@pytest.mark.parametrize("W,H,levels,radius,iters", [(160, 96, 3, 4, 3), (97, 61, 2, 2, 2), (64, 48, 1, 6, 4), (320, 180, 3, 4, 3),
                                                   (80, 60, 2, 3, 2), (50, 40, 1, 5, 1), (70, 50, 2, 9, 1),
                                                   (101, 101, 3, 4, 2)])   # pyramid + plane sizes odd: workspace segments need padding to stay 16-byte aligned   # odd radii: run-time-radius kernels
def test_lk_flow_bit_exact_vs_oracle(ctx, W, H, levels, radius, iters):
    """hip_lk == the build's CPU restatement, same bits (the algorithm itself is build-defined: the reference calls
    OpenCV's Farneback, cv-decoder/src/lib.rs:188-199 -> parity unpinned w.r.t. the reference)."""
    fr = synth.luma_sequence(2, W, H, max_step=3, seed=synth.SEED0 + W)
    f_o = oracle.lk_flow(fr[0], fr[1], levels, radius, iters)
    f_g, e_g = ctx.lk_flow(fr[0], fr[1], levels, radius, iters, want_entries=True)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    np.testing.assert_array_equal(e_g.view(np.uint32), oracle.flow_to_entries(f_o).view(np.uint32))   # cv-decoder records


@pytest.mark.parametrize("radius", [2, 4, 6, 3])
@pytest.mark.parametrize("W,H,levels", [(1, 1, 1), (2, 2, 3), (5, 3, 2), (31, 7, 3), (33, 9, 4), (40, 30, 8), (17, 200, 5), (200, 17, 5),
                                        (32, 8, 1), (64, 16, 2), (65, 17, 3)])
def test_lk_flow_degenerate_geometries(ctx, W, H, levels, radius):
    """Frames smaller than a tile, one-pixel pyramid levels (40 x 30 at 8 levels ends in 1 x 1), strips one tile wide or high,
    exact multiples of the tile: every window sample is clamped somewhere.  Same bits as the oracle, with and without records."""
    fr = synth.luma_sequence(2, max(W, 8), max(H, 8), max_step=1, seed=1000 + W + H)[:, :H, :W]
    fr = np.ascontiguousarray(fr)
    f_o = oracle.lk_flow(fr[0], fr[1], levels, radius, 2)
    f_g, e_g = ctx.lk_flow(fr[0], fr[1], levels, radius, 2, want_entries=True)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    np.testing.assert_array_equal(e_g.view(np.uint32), oracle.flow_to_entries(f_o).view(np.uint32))
    assert ctx.lk_helped_tiles() == 0


@pytest.mark.parametrize("fall_step", [0, 1, 2, 16 + 0, 16 + 1, 16 + 2, 32 + 0, 32 + 2, 48 + 1])
@pytest.mark.parametrize("radius", [2, 4, 6])
def test_lk_flow_hand_over_in_the_middle_of_a_level(hooks_ctx, fall_step, radius):
    """The level kernel keeps a tile's flow on chip across the Gauss-Newton steps of a level; a tile whose current-frame
    rectangle does not fit LDS at step k is finished inside the same launch: its pixels in groups around an anchor pixel's
    rectangle (up to 8 rounds), whatever is left by the oracle's per-sample form from global memory.  The test hook makes
    every other tile take that path at step k (low 4 bits: 0 = the level's first step, tensor sums included, 1 / 2 =
    mid-level) and limits the grouping rounds (bits 4..: 0 = default, 1 + n = n rounds -> 16 + k: every pixel through the
    leftover path, 32 + k / 48 + k: one / two rounds, then leftovers), with and without the records output: same bits as the
    oracle either way."""
    ctx = hooks_ctx                     # the hook is compiled only into libofps_hip_testhooks.so
    ctx.set_option("OFPS_HIP_LK_TEST_FALL", str(fall_step))
    W, H, levels, iters = 320, 180, 3, 3
    fr = synth.luma_sequence(2, W, H, max_step=3, seed=77 + radius)
    f_o = oracle.lk_flow(fr[0], fr[1], levels, radius, iters)
    f_g, e_g = ctx.lk_flow(fr[0], fr[1], levels, radius, iters, want_entries=True)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    np.testing.assert_array_equal(e_g.view(np.uint32), oracle.flow_to_entries(f_o).view(np.uint32))
    f_only = ctx.lk_flow(fr[0], fr[1], levels, radius, iters, want_entries=False)
    np.testing.assert_array_equal(f_only.view(np.uint32), f_o.view(np.uint32))
    # records only (no flow plane requested: the device-pointer entry point, what hip_lk's decode path calls)
    import torch
    dfr = torch.from_numpy(fr).cuda()
    d_ent = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    try:
        ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), W, H, W, levels, radius, iters, None, d_ent.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.use_own_stream()
    np.testing.assert_array_equal(d_ent.cpu().numpy().view(np.uint32), oracle.flow_to_entries(f_o).view(np.uint32))


def _cu_masked_stream(n_cus: int):
    """a HIP stream whose kernels may only use the first n_cus compute units (hipExtStreamCreateWithCUMask)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    words = (C.c_uint32 * 8)(*([0] * 8))
    for cu in range(n_cus):
        words[cu // 32] |= 1 << (cu % 32)
    stream = C.c_void_p(0)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, words)
    assert rc == 0, rc
    return hip, stream


@pytest.mark.parametrize("mask", ["first32", "every_other", "every_eighth"])
def test_almeida_one_xcd_cluster_on_cu_masked_streams(ctx, mask):
    """The one-XCD form of the small cluster solve rests on round-robin dispatch; on a stream whose kernels may only use some of the
    CUs the working workgroups may or may not share an XCD -- step 0 finds out from their XCC_IDs and the launch keeps the flat
    exchange if they do not.  Either way: the unmasked launch's bits, nobody finishes alone."""
    import ctypes as C
    e = synth.rotation_field(120, 67)
    q_ref, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
    np.testing.assert_allclose(q_ref, oracle.solve_ypr_given(e, oracle.camera(16 / 9, 22.275)), atol=2e-6, rtol=0)
    hip = C.CDLL("libamdhip64.so")
    words = (C.c_uint32 * 8)(*([0] * 8))
    cus = {"first32": range(32), "every_other": range(0, 256, 2), "every_eighth": range(0, 256, 8)}[mask]
    for cu in cus:
        words[cu // 32] |= 1 << (cu % 32)
    stream = C.c_void_p(0)
    assert hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, words) == 0
    rec0 = ctx.almeida_recoveries()
    try:
        ctx.set_stream(stream.value)
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster")
        for _ in range(3):
            q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
            np.testing.assert_array_equal(q.view(np.uint32), q_ref.view(np.uint32))
    finally:
        ctx.set_option("OFPS_HIP_ALMEIDA_PATH", None)
        ctx.use_own_stream()
        hip.hipStreamDestroy(stream)
    assert ctx.almeida_recoveries() == rec0


@pytest.mark.parametrize("mode", ["budget", "reversed", "permuted", "cu_mask", "cu_mask_permuted"])
def test_lk_forward_progress_does_not_depend_on_the_dispatcher(hooks_ctx, mode):
    """The pyramid is ONE launch in which a tile waits for its parent tile's flows; a tile whose parent has not published in time computes
    the missing ancestors ITSELF (lk_help_ancestors), so every workgroup finishes whatever the dispatcher does and no flow is made from
    an unfinished parent.  Hooks: a wait budget of one tick (nearly every tile computes its own ancestors), the blocks taking the
    positions in REVERSE (level 0 first, the coarsest level last) or in a pseudo-random permutation with random start delays; and a
    stream masked to 32 of the 256 CUs.  Every entry-point form returns the oracle's bits; the helped-tile count says the path ran."""
    ctx = hooks_ctx
    W, H, levels, radius, iters = 640, 360, 3, 4, 3
    fr = synth.luma_sequence(4, W, H, max_step=3, seed=311)
    f_o = oracle.lk_flow(fr[0], fr[1], levels, radius, iters)
    ctx.lk_reset()
    ref_dec = [ctx.lk_decode(fr[k], fr[k + 1], levels, radius, iters) for k in range(3)]     # undisturbed
    helped0 = ctx.lk_helped_tiles()
    hip = stream = None
    try:
        if mode == "budget":
            ctx.set_option("OFPS_HIP_LK_TEST_WAIT_BUDGET", "1")
        if mode in ("reversed",):
            ctx.set_option("OFPS_HIP_LK_TEST_ORDER", "1")
        if mode in ("permuted", "cu_mask_permuted"):
            ctx.set_option("OFPS_HIP_LK_TEST_ORDER", "2")
        if mode.startswith("cu_mask"):
            hip, stream = _cu_masked_stream(32)
            ctx.set_stream(stream.value)
        f_g, e_g = ctx.lk_flow(fr[0], fr[1], levels, radius, iters, want_entries=True)
        np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
        np.testing.assert_array_equal(e_g.view(np.uint32), oracle.flow_to_entries(f_o).view(np.uint32))
        if mode == "budget":
            assert ctx.lk_helped_tiles() > helped0                      # the path under test did run (the other modes need it only when the launch
                                                                        # exceeds the resident workgroups: the 1080p test below)
        for rad in (2, 6):                                               # the other tiled kernels
            np.testing.assert_array_equal(ctx.lk_flow(fr[0], fr[1], levels, rad, 2).view(np.uint32), oracle.lk_flow(fr[0], fr[1], levels, rad, 2).view(np.uint32))
        # a deeper pyramid: ancestors of ancestors
        np.testing.assert_array_equal(ctx.lk_flow(fr[1], fr[2], 5, radius, 2).view(np.uint32), oracle.lk_flow(fr[1], fr[2], 5, radius, 2).view(np.uint32))
        ent, grid = ctx.lk_decode(fr[0], fr[1], levels, radius, iters)
        assert grid == ref_dec[0][1]
        np.testing.assert_array_equal(ent.view(np.uint32), ref_dec[0][0].view(np.uint32))
        # synchronous stream form
        ctx.lk_reset()
        assert ctx.lk_push_frame(fr[0], levels, radius, iters) is None
        for k in range(1, 4):
            ent, grid = ctx.lk_push_frame(fr[k], levels, radius, iters)
            np.testing.assert_array_equal(ent.view(np.uint32), ref_dec[k - 1][0].view(np.uint32))
        # read-ahead form, two tickets in flight
        ctx.lk_reset()
        pins = [ctx.pinned_frame(H, W) for _ in range(4)]
        for k in range(4): np.copyto(pins[k], fr[k])
        tickets = [ctx.lk_push_frame_async(pins[0], levels, radius, iters), ctx.lk_push_frame_async(pins[1], levels, radius, iters)]
        got = [ctx.lk_frame_wait(tickets[0])]
        tickets.append(ctx.lk_push_frame_async(pins[2], levels, radius, iters))
        got.append(ctx.lk_frame_wait(tickets[1]))
        tickets.append(ctx.lk_push_frame_async(pins[3], levels, radius, iters))
        got.append(ctx.lk_frame_wait(tickets[2]))
        got.append(ctx.lk_frame_wait(tickets[3]))
        assert got[0] is None
        for k in range(1, 4):
            np.testing.assert_array_equal(got[k][0].view(np.uint32), ref_dec[k - 1][0].view(np.uint32))
        ctx.lk_reset()
        # the device-pointer entry point: nothing to repair afterwards, the sync is clean and the records are the oracle's
        if not mode.startswith("cu_mask"):
            import torch
            dfr = torch.from_numpy(fr[:2]).cuda()
            d_ent = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
            ctx.use_torch_stream()
            try:
                ctx.lk_flow_dev(dfr[0].data_ptr(), dfr[1].data_ptr(), W, H, W, levels, radius, iters, None, d_ent.data_ptr())
                ctx.sync()
            finally:
                ctx.use_own_stream()
            np.testing.assert_array_equal(d_ent.cpu().numpy().view(np.uint32), oracle.flow_to_entries(f_o).view(np.uint32))
    finally:
        ctx.set_option("OFPS_HIP_LK_TEST_WAIT_BUDGET", None)
        ctx.set_option("OFPS_HIP_LK_TEST_ORDER", None)
        ctx.use_own_stream()
        ctx.lk_reset()
        if stream is not None:
            hip.hipStreamDestroy(stream)
    # hooks off, whole device: an in-order launch needs no help
    before = ctx.lk_helped_tiles()
    f_g = ctx.lk_flow(fr[0], fr[1], levels, radius, iters)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    assert ctx.lk_helped_tiles() == before


def test_lk_forward_progress_1080p_reversed_positions(hooks_ctx):
    """BASELINE's size with the worst start order (level 0's 4,050 tiles before their parents): every level-0 tile computes its two
    ancestors itself -- three times the work, the same bits."""
    ctx = hooks_ctx
    fr = synth.luma_sequence(2, 1920, 1080, max_step=3, seed=11)
    f_o = oracle.lk_flow(fr[0], fr[1], 3, 4, 3)
    np.testing.assert_array_equal(ctx.lk_flow(fr[0], fr[1], 3, 4, 3).view(np.uint32), f_o.view(np.uint32))     # (allocates the flag buffer: the count restarts)
    assert ctx.lk_helped_tiles() == 0                                   # in order: nobody needs help
    ctx.set_option("OFPS_HIP_LK_TEST_ORDER", "1")
    try:
        f_g = ctx.lk_flow(fr[0], fr[1], 3, 4, 3)
        helped = ctx.lk_helped_tiles()
        print(f"1080p, positions reversed: {helped} ancestor tiles computed by waiting children (the pyramid has {272 + 1020} of them)")
        assert helped > 1000
    finally:
        ctx.set_option("OFPS_HIP_LK_TEST_ORDER", None)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


def test_lk_read_ahead_refuses_a_geometry_change_with_a_ticket_in_flight(ctx):
    """a push with a new geometry used to drain the stream, which marked the other ticket collected and lost its
    records; it is refused now (like the multi-device form), and a failed push leaves the stream position alone."""
    from ofps_amd.runtime import OfpsHipError
    fr = synth.luma_sequence(3, 320, 192, max_step=2, seed=9)
    small = np.ascontiguousarray(fr[2][:96, :160])
    ref = ctx.lk_decode(fr[0], fr[1], 3, 4, 2)
    ctx.lk_reset()
    t0 = ctx.lk_push_frame_async(fr[0], 3, 4, 2)
    t1 = ctx.lk_push_frame_async(fr[1], 3, 4, 2)
    assert ctx.lk_frame_wait(t0) is None
    with pytest.raises(OfpsHipError) as ei:
        ctx.lk_push_frame_async(small, 3, 4, 2)                                  # t1 is in flight
    assert ei.value.code == -1 and "in flight" in str(ei.value)
    got = ctx.lk_frame_wait(t1)                                                  # still collectable, records intact
    np.testing.assert_array_equal(got[0].view(np.uint32), ref[0].view(np.uint32))
    t2 = ctx.lk_push_frame_async(small, 3, 4, 2)                                 # nothing in flight: the new geometry restarts the stream
    assert ctx.lk_frame_wait(t2) is None
    ctx.lk_reset()


def test_lk_tile_flags_are_nobodys_scratch(ctx):
    """The one-launch pyramid tags its tile flags with a per-call epoch and never clears them, so the flag buffer must not be
    anybody else's workspace: in round 4 it shared a slot with the densifier's per-cell begin[] table, whose values (record
    indices: small integers) would have read as "parent tile done" for the launch whose epoch they equal.  Densify calls whose
    begin[] covers 0..n interleaved with flows for a run of epochs: every flow has the oracle's bits."""
    W, H = 160, 96
    fr = synth.luma_sequence(2, W, H, max_step=2, seed=5)
    f_o = oracle.lk_flow(fr[0], fr[1], 3, 4, 2)
    rng = np.random.default_rng(3)
    for k in range(24):
        n = 40 + 7 * k                                       # begin[] of a 16 x 9 field then holds most integers below n
        e = np.zeros((n, 4), np.float32)
        e[:, 0] = (np.arange(n) % 16 + 0.5) / 16; e[:, 1] = (np.arange(n) // 16 % 9 + 0.5) / 9
        e[:, 2:] = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        ctx.densify(e, 16, 9)
        f_g = ctx.lk_flow(fr[0], fr[1], 3, 4, 2)
        np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    assert ctx.lk_helped_tiles() == 0


@pytest.mark.parametrize("radius", [2, 4, 6])
def test_lk_one_launch_per_level_gives_the_same_bits(ctx, radius):
    """OFPS_HIP_LK_SERIAL=1: the pyramid level by level (what the repair above runs) -- same bits as the one-launch form."""
    W, H, levels, iters = 322, 181, 3, 2
    fr = synth.luma_sequence(2, W, H, max_step=3, seed=17 + radius)
    f_one, e_one = ctx.lk_flow(fr[0], fr[1], levels, radius, iters, want_entries=True)
    ctx.set_option("OFPS_HIP_LK_SERIAL", "1")
    try:
        f_ser, e_ser = ctx.lk_flow(fr[0], fr[1], levels, radius, iters, want_entries=True)
    finally:
        ctx.set_option("OFPS_HIP_LK_SERIAL", None)
    np.testing.assert_array_equal(f_ser.view(np.uint32), f_one.view(np.uint32))
    np.testing.assert_array_equal(e_ser.view(np.uint32), e_one.view(np.uint32))
    np.testing.assert_array_equal(f_ser.view(np.uint32), oracle.lk_flow(fr[0], fr[1], levels, radius, iters).view(np.uint32))
    assert ctx.lk_helped_tiles() == 0


def test_lk_flow_on_a_smooth_subpixel_camera_warp(ctx):
    """Sub-pixel flows that vary smoothly over the frame (roll + zoom + shift): sample origins flip along lines instead of per
    region, every lane of a tile has its own fractions.  Same bits as the oracle; the recovered flow is the planted field."""
    W, H = 384, 216
    fr = synth.camera_warp_pair(W, H, roll_deg=0.6, zoom=1.006, shift=(2.3, -1.6), seed=3)
    f_o = oracle.lk_flow(fr[0], fr[1], 3, 4, 3)
    f_g = ctx.lk_flow(fr[0], fr[1], 3, 4, 3)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    a, cx, cy = np.deg2rad(0.6), (W - 1) / 2, (H - 1) / 2
    ex = cx + 1.006 * (np.cos(a) * (xx - cx) - np.sin(a) * (yy - cy)) + 2.3 - xx      # prev(x, y) ~ cur(x + u, y + v): the inverse of the warp
    ey = cy + 1.006 * (np.sin(a) * (xx - cx) + np.cos(a) * (yy - cy)) - 1.6 - yy
    inner = np.s_[24:-24, 24:-24]
    assert np.abs(f_g[..., 0][inner] + ex[inner]).mean() < 0.25 and np.abs(f_g[..., 1][inner] + ey[inner]).mean() < 0.25


def test_lk_flow_recovers_planted_translation(ctx):
    base = synth.luma_sequence(1, 640 + 64, 360 + 64, max_step=0, noise=0, seed=5)[0]
    dx, dy = 5, -3
    prev = np.ascontiguousarray(base[32:32 + 360, 32:32 + 640])
    cur = np.ascontiguousarray(base[32 - dy:32 - dy + 360, 32 - dx:32 - dx + 640])      # cur(x+dx, y+dy) = prev(x, y)
    f = ctx.lk_flow(prev, cur, 3, 4, 3)
    inner = f[32:-32, 32:-32]
    assert np.abs(inner - np.array([dx, dy], np.float32)).mean() < 1e-3


@pytest.mark.parametrize("W,H,dx,levels", [(1030, 40, 5, 1), (1030, 48, 5, 3), (1027, 40, 4, 2), (518, 64, 6, 2), (2053, 40, 5, 1),
                                           (1030, 40, -5, 1)])
def test_lk_flow_width_just_above_a_power_of_two_with_flows_leaving_the_right_edge(ctx, W, H, dx, levels):
    """ADVICE r3 (lk.hip fast column origins): a planted translation of dx pixels makes the sums x + k - r + u of the right-most
    interior tile straddle the power of two just below W with fractions within rounding of an integer, and push the last
    sample origin onto / past the frame's right edge where lk_origin clamps it: windows whose floors skip one across the
    binade must not pass for consecutive.  Same bits as the oracle."""
    base = synth.luma_sequence(1, W + 64, H + 64, max_step=0, noise=0, seed=9 + W)[0]
    prev = np.ascontiguousarray(base[32:32 + H, 32:32 + W])
    cur = np.ascontiguousarray(base[32:32 + H, 32 - dx:32 - dx + W])                  # cur(x + dx, y) = prev(x, y)
    f_o = oracle.lk_flow(prev, cur, levels, 4, 3)
    f_g = ctx.lk_flow(prev, cur, levels, 4, 3)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))
    if levels >= 2:
        assert np.abs(f_o[8:-8, 16:-16, 0] - dx).mean() < 0.5                          # the flows do point where the case needs them


def test_lk_spec_revision_of_the_library_is_the_oracles(ctx):
    """OFPS_LK_SPEC_FMA (lk.hip) and ORC_LK_SPEC_FMA (oracle) are independent build switches: they must be set alike, or
    every bit-exact LK test compares two different specs."""
    assert ctx.lk_spec_revision() == oracle.lk_spec_revision() == 2


def _crafted_binade_init(W, H):
    """Starting flow (one level) whose last-column pixels of the right-most interior tile have window floors that skip one
    across the power of two below W while the LAST floor is clamped to W -- the input of ADVICE r3's counterexample -- and
    every other pixel of the tile has consecutive floors, so nothing else in the workgroup vetoes the fast path."""
    R = 4
    init = np.zeros((H, W, 2), np.float32)
    init[..., 0] = 2.0
    p2 = 1 << (W.bit_length() - 1)
    assert p2 < W <= p2 + 2 * R + 1 and (W - 36) % 32 == 0       # tile x0 = W - 36 is interior and its last pixel is p2 - 1 (W = p2 + 4)
    xs = p2 - 1
    init[:, xs, 0] = np.float32(2.0) - np.float32(4.6e-5)        # (with exactly 2.0 the clamped last floor repeats: end difference 2r - 1, vetoed)
    k = np.arange(2 * R + 1)
    x = np.arange(W)[None, :].repeat(H, 0)
    fl = np.floor((x[..., None] + k - R).astype(np.float32) + init[..., 0:1])
    cl = np.clip(fl, -1, W)
    unguarded = (cl[..., 0] >= 0) & (cl[..., -1] - cl[..., 0] == 2 * R)
    truth = (np.diff(cl, axis=-1) == 1).all(-1)
    tile = np.s_[:, W - 36:W - 4]
    assert unguarded[tile].all() and not truth[tile].all()       # round 3's test passes the tile, the floors are not consecutive
    return init


@pytest.mark.parametrize("iters", [1, 3])
def test_lk_flow_clamped_last_origin_across_a_binade_is_not_consecutive(ctx, iters):
    """The deterministic form of ADVICE r3's counterexample (lk.hip:812): W = 1028, r = 4, flows 2 - 4.6e-5 in column 1023:
    floors 1020 .. 1023, 1025 .. 1029 with the last clamped to W -- end difference 2r although a floor is skipped.  The
    starting flow goes in through ofps_hip_lk_flow_init_dev (oracle: orc_lk_flow_init); same bits."""
    W, H = 1028, 16
    fr = synth.luma_sequence(2, W, H, max_step=2, seed=31)
    init = _crafted_binade_init(W, H)
    f_o = oracle.lk_flow(fr[0], fr[1], 1, 4, iters, init=init)
    f_g = ctx.lk_flow_init(fr[0], fr[1], 1, 4, iters, init)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


@pytest.mark.parametrize("W,H,levels,radius,amp,seed", [(200, 120, 1, 4, 30.0, 1), (200, 120, 3, 4, 12.0, 2), (333, 77, 2, 4, 40.0, 3),
                                                        (1030, 24, 1, 4, 8.0, 4), (160, 96, 2, 2, 25.0, 5), (160, 96, 2, 6, 25.0, 6),
                                                        (90, 70, 2, 3, 10.0, 7), (520, 40, 2, 4, 600.0, 8)])
def test_lk_flow_with_wild_starting_flows(ctx, W, H, levels, radius, amp, seed):
    """Starting flows no pyramid would produce -- per-pixel noise of +-amp pixels on top of region jumps, rows and columns of
    exact integers and of values one ulp below them, flows far outside the frame -- put every tile of the coarsest level
    through the rectangle-does-not-fit path, the per-column origins, the clamped rows and the row-reuse exceptions.  Same
    bits as the oracle."""
    rng = np.random.default_rng(seed)
    fr = synth.luma_sequence(2, W, H, max_step=3, seed=50 + seed)
    h, w = oracle.lk_coarsest_shape(W, H, levels)
    init = rng.uniform(-amp, amp, (h, w, 2)).astype(np.float32)
    init[: h // 3] = np.round(init[: h // 3])                                        # integers: fractions exactly 0
    init[h // 3: h // 2] = np.nextafter(np.round(init[h // 3: h // 2]), np.float32(-np.inf)).astype(np.float32)
    init[:, : w // 4] = init[0, 0]                                                   # a coherent region (the staged path)
    init[:, w // 4: w // 2] *= np.float32(0.02)                                      # ordinary sub-pixel flows
    f_o = oracle.lk_flow(fr[0], fr[1], levels, radius, 2, init=init)
    f_g = ctx.lk_flow_init(fr[0], fr[1], levels, radius, 2, init)
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


def test_lk_to_rotation_end_to_end(ctx):
    """cfg3 shape at reduced size: frames -> per-pixel flow -> cv-decoder records -> densify 150x84 -> Almeida LSQ;
    every stage equals the oracle run stage by stage."""
    fr = synth.luma_sequence(2, 480, 272, max_step=2, seed=77)
    f_g, e_g = ctx.lk_flow(fr[0], fr[1], 3, 4, 3, want_entries=True)
    e_o = oracle.flow_to_entries(oracle.lk_flow(fr[0], fr[1], 3, 4, 3))
    np.testing.assert_array_equal(e_g.view(np.uint32), e_o.view(np.uint32))
    d_g, d_o = ctx.densify_to_entries(e_g, 150, 84), oracle.densify_to_entries(e_o, 150, 84)
    np.testing.assert_array_equal(d_g.view(np.uint32), d_o.view(np.uint32))
    q_g, _ = ctx.almeida(d_g, 16 / 9, 22.275, use_ransac=False)
    np.testing.assert_allclose(q_g, oracle.solve_ypr_given(d_o, oracle.camera(16 / 9, 22.275)), atol=2e-6, rtol=0)


# ------------------------------------------------------------------ argument validation / degenerate sizes
def test_degenerate_geometries_and_bad_arguments(ctx):
    from ofps_amd.runtime import OfpsHipError
    # frame smaller than one block: zero vectors, not an error
    tiny = np.zeros((8, 8), np.uint8)
    assert len(ctx.sad_flow(tiny, tiny, 16, 16)) == 0
    # a single block, search window entirely clipped except (0,0)
    one = synth.random_luma(2, 16, 16, seed=3)
    ent, best = ctx.sad_flow(one[0], one[1], 16, 16, want_best=True)
    _, best_o = oracle.sad_flow(one[0], one[1], 16, 16)
    np.testing.assert_array_equal(best, best_o)
    assert tuple(best[0][:2]) == (0, 0)
    # densify / detect / almeida reject nonsense loudly
    e = _entries(10, 1)
    with pytest.raises(OfpsHipError):
        ctx.densify(e, 0, 4)
    with pytest.raises(OfpsHipError):
        ctx.densify(e, 300, 300)                      # > 65536 cells
    with pytest.raises(OfpsHipError):
        ctx.detect(e, min_size=1e-9, subdivide=16)    # block_dim far beyond the 160 the properties allow
    with pytest.raises(OfpsHipError):
        ctx.almeida(e, -1.0, 90.0)
    with pytest.raises(OfpsHipError):
        ctx.lk_flow(tiny, tiny, levels=0)
    with pytest.raises(OfpsHipError):
        ctx.set_sad_mode(7)
    # the context stays usable after errors
    assert ctx.densify(e, 4, 4).shape == (4, 4, 2)


def test_lk_decode_matches_cv_decoder_output_stage(ctx):
    """hip_lk process_frame body: flow -> records -> densifier down-sampling to the capped grid (150x150 -> 150x84
    at 16:9, cv-decoder/src/lib.rs:98-121) -> visited cells in (x,y) order; equals the oracle stage by stage."""
    fr = synth.luma_sequence(2, 480, 270, max_step=2, seed=9)
    ent, (gw, gh) = ctx.lk_decode(fr[0], fr[1])
    assert (gw, gh) == (150, 84)
    e_o = oracle.densify_to_entries(oracle.flow_to_entries(oracle.lk_flow(fr[0], fr[1], 3, 4, 3)), 150, 84)
    np.testing.assert_array_equal(ent.view(np.uint32), e_o.view(np.uint32))
    # small frame: the cap is the frame itself (min(w, cols), min(h, rows))
    fr = synth.luma_sequence(2, 96, 64, max_step=1, seed=2)
    ent, (gw, gh) = ctx.lk_decode(fr[0], fr[1])
    assert (gw, gh) == (96, 64) and len(ent) == 96 * 64


# ------------------------------------------------------------------ cv-decoder contrast mask (rank-4 "next" row)
@pytest.mark.parametrize("W,H", [(1, 1), (3, 2), (7, 5), (64, 16), (65, 17), (333, 77), (640, 360)])
def test_contrast_mask_is_bit_exact(ctx, W, H):
    """Sobel(1,1,k5) > 20 -> dilate(ellipse 11x11) (cv-decoder/src/lib.rs:203-237): integer-exact vs both restatements,
    tile seams, frames smaller than the halo, reflect-101 borders."""
    from oracle import np_oracle
    g = synth.flatten_regions(synth.luma_sequence(1, W, H, max_step=0, seed=W * 31 + H), region=24, seed=W + H)[0]
    m = ctx.contrast_mask(g)
    np.testing.assert_array_equal(m, oracle.contrast_mask(g))
    np.testing.assert_array_equal(m, np_oracle.contrast_mask(g))
    if W >= 64 and H >= 16:
        assert 0 < m.mean() < 1
    r = synth.random_luma(1, W, H, seed=5)[0]
    np.testing.assert_array_equal(ctx.contrast_mask(r), oracle.contrast_mask(r))
    z = np.full((H, W), 200, np.uint8)
    assert not ctx.contrast_mask(z).any()


def test_contrast_mask_device_resident_1080p(ctx):
    import torch
    g = synth.flatten_regions(synth.luma_sequence(1, 1920, 1080, max_step=0, seed=77), region=96, seed=3)[0]
    d_g = torch.from_numpy(g).cuda()
    d_m = torch.zeros((1080, 1920), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.contrast_mask_dev(d_g.data_ptr(), 1920, 1080, 1920, d_m.data_ptr())
    ctx.sync()
    np.testing.assert_array_equal(d_m.cpu().numpy(), oracle.contrast_mask(g))


def test_lk_decode_with_contrast_mask(ctx):
    """The masked record loop (cv-decoder/src/lib.rs:251-291): masked pixels contribute nothing, survivors keep raster
    order (which fixes the densifier's f32 summation order) -- bit-exact per stage, in both output modes."""
    fr = synth.flatten_regions(synth.luma_sequence(2, 480, 270, max_step=2, seed=12), region=40, seed=8)
    flow = oracle.lk_flow(fr[0], fr[1], 3, 4, 3)
    mask = oracle.contrast_mask(fr[1])
    assert 0.1 < mask.mean() < 0.95
    rec_o = oracle.masked_flow_to_entries(flow, mask)
    ent, (gw, gh) = ctx.lk_decode(fr[0], fr[1], contrast_mask=True, fullres_records=True)
    assert (gw, gh) == (480, 270)
    np.testing.assert_array_equal(ent.view(np.uint32), rec_o.view(np.uint32))
    ent, (gw, gh) = ctx.lk_decode(fr[0], fr[1], contrast_mask=True)
    assert (gw, gh) == (150, 84)
    e_o = oracle.densify_to_entries(rec_o, 150, 84)
    assert len(e_o) < 150 * 84                         # some cells are never visited
    np.testing.assert_array_equal(ent.view(np.uint32), e_o.view(np.uint32))
    # per-pixel without the mask = every pixel
    ent, _ = ctx.lk_decode(fr[0], fr[1], fullres_records=True)
    np.testing.assert_array_equal(ent.view(np.uint32), oracle.flow_to_entries(flow).view(np.uint32))
    # nothing survives a flat frame: zero records, not an error
    flat = np.full((2, 64, 96), 90, np.uint8)
    ent, _ = ctx.lk_decode(flat[0], flat[1], contrast_mask=True)
    assert len(ent) == 0


def test_lk_push_frame_stream_matches_pairwise_decode(ctx):
    """ofps_hip_lk_push_frame keeps the previous frame on the device: over a sequence it must return what the stateless
    ofps_hip_lk_decode returns for each consecutive pair (masked and unmasked, down-sampled and per pixel), nothing for
    the first frame, and start over after a reset or a geometry change."""
    fr = synth.flatten_regions(synth.luma_sequence(5, 320, 180, max_step=2, seed=21), region=40, seed=3)
    for kw in (dict(), dict(contrast_mask=True), dict(fullres_records=True), dict(contrast_mask=True, fullres_records=True)):
        ctx.lk_reset()
        assert ctx.lk_push_frame(fr[0], **kw) is None
        for k in range(1, 5):
            ent, grid = ctx.lk_push_frame(fr[k], **kw)
            ent_o, grid_o = ctx.lk_decode(fr[k - 1], fr[k], **kw)            # stateless: does not touch the stream's frames
            assert grid == grid_o
            np.testing.assert_array_equal(ent.view(np.uint32), ent_o.view(np.uint32))
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[0]) is None
    assert ctx.lk_push_frame(fr[1]) is not None
    assert ctx.lk_push_frame(fr[0][:90, :160].copy()) is None                # geometry change
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[2]) is None


@pytest.mark.parametrize("kw", [dict(), dict(contrast_mask=True), dict(fullres_records=True), dict(fullres_records=True, contrast_mask=True)])
def test_lk_push_frame_async_equals_the_synchronous_decoder(ctx, kw):
    """The ticketed read-ahead form (ofps_hip_lk_push_frame_async / _frame_wait, two tickets in flight: the upload of frame k+1
    on the copy stream beside the flow of pair (k-1, k), the records written by the ticket's last kernel straight into its
    page-locked block) returns, frame by frame, exactly what ofps_hip_lk_decode returns for the pair -- every output mode;
    cv-decoder/src/lib.rs:82-158 is the loop it serves."""
    W, H, F = 320, 180, 7
    fr = synth.flatten_regions(synth.luma_sequence(F, W, H, max_step=3, seed=synth.SEED0 + 21), region=48, seed=3)
    pins = [ctx.pinned_frame(H, W) for _ in range(3)]
    ctx.lk_reset()
    got, prev = [], None
    for k in range(F):                                     # push k, then collect k - 1: two tickets in flight
        np.copyto(pins[k % 3], fr[k])
        t = ctx.lk_push_frame_async(pins[k % 3], max_w=60, max_h=60, **kw)
        if prev is not None:
            got.append(ctx.lk_frame_wait(prev))
        prev = t
    got.append(ctx.lk_frame_wait(prev))
    assert got[0] is None                                  # the first frame of a stream yields no vectors
    for k in range(1, F):
        want, grid = ctx.lk_decode(fr[k - 1], fr[k], max_w=60, max_h=60, **kw)
        assert got[k][1] == grid
        np.testing.assert_array_equal(got[k][0].view(np.uint32), want.view(np.uint32))
    # the synchronous form shares the stream: it continues it, and refuses to jump a ticket in flight
    r = ctx.lk_push_frame(fr[0], max_w=60, max_h=60, **kw)
    want, _ = ctx.lk_decode(fr[F - 1], fr[0], max_w=60, max_h=60, **kw)
    np.testing.assert_array_equal(r[0].view(np.uint32), want.view(np.uint32))
    t = ctx.lk_push_frame_async(pins[0], max_w=60, max_h=60, **kw)
    with pytest.raises(OfpsHipError):
        ctx.lk_push_frame(fr[1], max_w=60, max_h=60, **kw)
    ctx.lk_frame_wait(t)
    with pytest.raises(OfpsHipError):
        ctx.lk_frame_wait(t)                                 # collected already
    t0 = ctx.lk_push_frame_async(pins[1], max_w=60, max_h=60, **kw)
    t1 = ctx.lk_push_frame_async(pins[2], max_w=60, max_h=60, **kw)
    with pytest.raises(OfpsHipError):
        ctx.lk_push_frame_async(pins[0], max_w=60, max_h=60, **kw)      # a third frame needs the first ticket collected
    ctx.lk_frame_wait(t0); ctx.lk_frame_wait(t1)
    ctx.lk_reset()
    for p in pins:
        ctx.free_pinned(p)


def test_lk_push_frame_async_1080p_default_grid(ctx):
    """cfg3's shape: 1080p frames, the default 150 x 150 cap (150 x 84 cells), pageable AND page-locked sources."""
    W, H = 1920, 1080
    fr = synth.luma_sequence(4, W, H, max_step=3, seed=synth.SEED0 + 22)
    ctx.lk_reset()
    prev, got = None, []
    for k in range(4):
        t = ctx.lk_push_frame_async(fr[k])                 # pageable source: the copy is synchronous, the results are the same
        if prev is not None:
            got.append(ctx.lk_frame_wait(prev))
        prev = t
    got.append(ctx.lk_frame_wait(prev))
    for k in range(1, 4):
        want, grid = ctx.lk_decode(fr[k - 1], fr[k])
        assert grid == (150, 84) and got[k][1] == grid
        np.testing.assert_array_equal(got[k][0].view(np.uint32), want.view(np.uint32))
    ctx.lk_reset()


def test_lk_push_frame_stream_survives_every_other_entry_point(ctx):
    """The stream's previous frame lives in a device slot of its own: stateless calls that stage LARGER frames (they used to
    share -- and reallocate -- the slot) between two pushes leave the stream intact: the second push returns the vectors
    of (frame 0, frame 1), not of whatever the other call left behind."""
    fr = synth.flatten_regions(synth.luma_sequence(3, 320, 180, max_step=2, seed=22), region=40, seed=4)
    big = synth.luma_sequence(2, 640, 360, max_step=8, seed=23)
    want = {k: ctx.lk_decode(fr[k - 1], fr[k]) for k in (1, 2)}
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[0]) is None
    ctx.sad_flow(big[0], big[1], 16, 8)                                      # S_FRAMES regrown and overwritten
    ctx.lk_flow(big[0], big[1], 3, 4, 3)
    ctx.contrast_mask(big[1])
    ctx.lk_decode(big[0], big[1])
    ent, grid = ctx.lk_push_frame(fr[1])
    assert grid == want[1][1]
    np.testing.assert_array_equal(ent.view(np.uint32), want[1][0].view(np.uint32))
    ctx.lk_decode(big[1], big[0], fullres_records=True)
    ent, grid = ctx.lk_push_frame(fr[2])
    assert grid == want[2][1]
    np.testing.assert_array_equal(ent.view(np.uint32), want[2][0].view(np.uint32))


@pytest.mark.parametrize("W,H,gw,gh", [(480, 270, 150, 84), (321, 123, 150, 57), (97, 61, 14, 14), (64, 48, 64, 48), (40, 30, 150, 84),
                                       (1920, 1080, 150, 84)])
@pytest.mark.parametrize("masked", [False, True])
def test_densify_raster_matches_generic_densify_and_oracle(ctx, W, H, gw, gh, masked):
    """ofps_hip_densify_raster_dev walks each cell's rectangle of pixels instead of sorting the records; on a per-pixel
    producer's records (with or without cv-decoder's mask) it must return the bits of the generic densifier and of the
    oracle's add_vector loop on the (compacted) records.  Grids finer than the frame (40x30 -> 150x84) leave cells
    empty; odd sizes put the cell boundaries at irregular pixel columns."""
    import torch
    rng = np.random.default_rng(W * 7 + H)
    flow = (rng.standard_normal((H, W, 2)) * 3).astype(np.float32)
    ent = oracle.flow_to_entries(flow)                                        # the producer's record expression
    mask = (rng.random((H, W)) < 0.6).astype(np.uint8) if masked else None
    rec = ent[mask.reshape(-1) != 0] if masked else ent
    f_o = oracle.densify(rec, gw, gh)
    d_ent = torch.from_numpy(ent).cuda()
    d_mask = torch.from_numpy(mask).cuda() if masked else None
    d_f = torch.full((gh, gw, 2), -7.0, dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    try:
        ctx.densify_raster_dev(d_ent.data_ptr(), d_mask.data_ptr() if masked else None, W, H, gw, gh, d_f.data_ptr(), verify=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_f.cpu().numpy().view(np.uint32), f_o.view(np.uint32))
        d_rec = torch.from_numpy(np.ascontiguousarray(rec)).cuda()
        d_g = torch.empty_like(d_f)
        ctx.densify_dev(d_rec.data_ptr(), len(rec), 1, gw, gh, d_g.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_f.cpu().numpy().view(np.uint32), d_g.cpu().numpy().view(np.uint32))
    finally:
        ctx.use_own_stream()


def test_densify_raster_verify_rejects_records_that_are_not_the_lattice(ctx):
    import torch
    from ofps_amd.runtime import OfpsHipError as HipError
    W, H = 64, 48
    ent = oracle.flow_to_entries(np.zeros((H, W, 2), np.float32))
    bad = ent.copy(); bad[100, 0] += 0.25                                     # one record somewhere else
    swapped = ent.copy(); swapped[[5, 6]] = swapped[[6, 5]]                    # same cells, not the raster order
    d_f = torch.zeros((14, 14, 2), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    try:
        for arr in (bad, swapped):
            d = torch.from_numpy(arr).cuda()
            with pytest.raises(HipError):
                ctx.densify_raster_dev(d.data_ptr(), None, W, H, 14, 14, d_f.data_ptr(), verify=True)
        d = torch.from_numpy(ent).cuda()
        ctx.densify_raster_dev(d.data_ptr(), None, W, H, 14, 14, d_f.data_ptr(), verify=True)
        with pytest.raises(HipError):
            ctx.densify_raster_dev(d.data_ptr(), None, W, H, 300, 300, d_f.data_ptr())      # > 65536 cells
    finally:
        ctx.use_own_stream()


# ------------------------------------------------------------------ a host without its own HIP binding (the Rust shim's view)
def test_resident_chain_through_the_c_abi_memory_plumbing():
    """ofps_hip_malloc / memcpy_h2d / *_dev / memcpy_d2h / timer on the context's own stream, no torch anywhere: frames
    go up once, SAD -> detect -> Almeida run on device-resident vectors, only the small results come back."""
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    try:
        W, H, B, R, F = 640, 368, 16, 16, 3
        fr = synth.luma_sequence(F, W, H, max_step=8, seed=33)
        nb = (W // B) * (H // B)
        dim = c.block_dim(0.05, 3)
        d_fr, d_ent = c.malloc(fr.nbytes), c.malloc((F - 1) * nb * 16)
        d_res, d_fld, d_q = c.malloc((F - 1) * 16), c.malloc((F - 1) * dim * dim * 8), c.malloc((F - 1) * 16)
        assert c.get_stream() != 0
        c.memcpy_h2d(d_fr, fr)
        c.timer_start()
        c.sad_flow_dev(d_fr, F, W, H, W, W * H, 0, B, R, d_ent, None)
        c.detect_dev(d_ent, nb, F - 1, 0.05, 3, 0.003, d_res, d_fld)
        c.almeida_dev(d_ent, nb, F - 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, d_q)
        ms = c.timer_stop()
        assert 0.0 < ms < 1000.0
        ent = np.zeros((F - 1, nb, 4), np.float32); res = np.zeros((F - 1, 4), np.int32)
        fld = np.zeros((F - 1, dim, dim, 2), np.float32); q = np.zeros((F - 1, 4), np.float32)
        c.memcpy_d2h(ent, d_ent); c.memcpy_d2h(res, d_res); c.memcpy_d2h(fld, d_fld); c.memcpy_d2h(q, d_q)
        cam = oracle.camera(16 / 9, 22.275)
        for k in range(F - 1):
            eo, _ = oracle.sad_flow(fr[k], fr[k + 1], B, R)
            np.testing.assert_array_equal(ent[k].view(np.uint32), eo.view(np.uint32))
            ro = oracle.detect_motion(eo)
            assert bool(res[k, 0]) == (ro is not None)
            if ro is not None:
                assert res[k, 1] == ro[0] and res[k, 2] == dim
                np.testing.assert_array_equal(fld[k].view(np.uint32), ro[1].view(np.uint32))
            np.testing.assert_allclose(q[k], oracle.solve_ypr_given(eo, cam), atol=2e-6, rtol=0)
        for p in (d_fr, d_ent, d_res, d_fld, d_q):
            c.free(p)
    finally:
        c.close()


def test_almeida_ransac_large_sample_count(ctx):
    """"Ransac samples" goes up to 16000 (almeida-estimator/src/lib.rs:92-95): a refit set above 8192 inliers cannot use
    the one-workgroup solver and takes the multi-launch path after one small read-back of the inlier count."""
    e = synth.rotation_field(200, 100, outlier_frac=0.1, seed=4)
    cam = oracle.camera(16 / 9, 39.6 * 9 / 16)
    q_g, _ = ctx.almeida(e, 16 / 9, 39.6 * 9 / 16, use_ransac=True, num_iters=50, inlier_deg=0.05, num_samples=12000, seed=9)
    q_o = oracle.solve_ypr_ransac(e, cam, 50, 0.05, 12000, seed=9)
    np.testing.assert_allclose(q_g, q_o, atol=1e-4, rtol=0)


def test_sad_pruned_equals_exhaustive_on_random_geometries_and_content(ctx):
    """30 seeded random cases (frame size, content, noise, motion): the pruned search must return exactly what the
    exhaustive kernel returns -- GPU vs GPU, so large counts are cheap; the exhaustive kernel itself is pinned by the
    oracle cases above."""
    rng = np.random.default_rng(2024)
    for case in range(30):
        W = int(rng.integers(3, 45)) * 16 + int(rng.integers(0, 16))
        H = int(rng.integers(2, 30)) * 16 + int(rng.integers(0, 16))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            fr = synth.luma_sequence(2, W, H, max_step=int(rng.integers(0, 17)), seed=case, region=1 << 14, noise=int(rng.integers(0, 4)))
        elif kind == 1:
            fr = synth.luma_sequence(2, W, H, max_step=16, seed=case, region=int(rng.choice([32, 64, 128])))
        elif kind == 2:
            fr = synth.random_luma(2, W, H, seed=case)
        elif kind == 3:
            fr = synth.flatten_regions(synth.luma_sequence(2, W, H, max_step=8, seed=case), region=48, seed=case)
        else:                                              # saturated / two-level content: many exact ties
            fr = (synth.random_luma(2, W, H, seed=case) > 200).astype(np.uint8) * 255
        _, b0 = ctx.sad_flow(fr[0], fr[1], 16, 16, want_best=True)
        ctx.set_sad_mode(ctx.SAD_PRUNED)
        try:
            _, b1 = ctx.sad_flow(fr[0], fr[1], 16, 16, want_best=True)
        finally:
            ctx.set_sad_mode(ctx.SAD_EXHAUSTIVE)
        np.testing.assert_array_equal(b1, b0, err_msg=f"case {case}: {W}x{H} kind {kind}")


def test_contexts_are_independent_across_host_threads():
    """Plugins are `Send`, not `Sync` (ofps/src/plugins/mod.rs:244,261,278): one context per plugin object, used by one
    thread at a time, different contexts concurrently from different threads (the tracking worker runs its estimators
    on a rayon pool, tracking/worker.rs:347-361).  Four threads, four contexts, interleaved calls: every result exact."""
    import threading
    from ofps_amd.runtime import HipContext
    results, errors = {}, []

    def work(tag):
        try:
            c = HipContext(0)
            fr = synth.luma_sequence(3, 320, 192, max_step=8, seed=100 + tag)
            outs = []
            for k in (1, 2):
                for _ in range(10):
                    ent = c.sad_flow(fr[k - 1], fr[k], 16, 8)
                    q, _ = c.almeida(ent, 16 / 9, 22.275, use_ransac=bool(tag & 1), num_iters=50, seed=7)
                    det = c.detect(ent)
                outs.append((ent, q, det))
            results[tag] = (fr, outs)
            c.close()
        except Exception as e:                  # surfaced below: an exception in a thread would otherwise be lost
            errors.append((tag, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    cam = oracle.camera(16 / 9, 22.275)
    for tag, (fr, outs) in results.items():
        for k, (ent, q, det) in zip((1, 2), outs):
            eo, _ = oracle.sad_flow(fr[k - 1], fr[k], 16, 8)
            np.testing.assert_array_equal(ent.view(np.uint32), eo.view(np.uint32))
            qo = oracle.solve_ypr_ransac(eo, cam, 50, 0.05, 1000, seed=7) if tag & 1 else oracle.solve_ypr_given(eo, cam)
            np.testing.assert_allclose(q, qo, atol=1e-4 if tag & 1 else 2e-6, rtol=0)
            do = oracle.detect_motion(eo)
            assert (det is None) == (do is None)
            if det is not None:
                assert det[0] == do[0]
                np.testing.assert_array_equal(det[1].view(np.uint32), do[1].view(np.uint32))      # the whole field, bit for bit
