"""hip_flow: Farneback's dense flow on the GPU (ofps_amd/csrc/farneback.hip) against its CPU restatement (oracle/farneback_oracle.c),
which restates the published algorithm in the form the reference's cv-decoder gets from OpenCV (cv-decoder/src/lib.rs:188-199).
PARITY UNPINNED: OpenCV is neither part of the reference tree nor installed; tools/external_parity/opencv_compare.py is the check
anyone with cv2 can run.  HIP vs the restatement: every stage uses the same operations in the same order, so IDENTITY is what is asserted
(tol = 0: README / DESIGN claim "bit-identical"); north_star's 1e-4 px is the bound against the reference's own arithmetic."""
import numpy as np
import pytest

import oracle
from ofps_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _check(f_g, f_o, tol=0.0):
    d = np.abs(f_g - f_o)
    assert np.isfinite(f_g).all()
    assert d.max() <= tol, (float(d.max()), np.unravel_index(d.argmax(), d.shape))
    return float(d.max()), float((f_g.view(np.uint32) == f_o.view(np.uint32)).mean())


@pytest.mark.parametrize("W,H", [(640, 360), (322, 181), (97, 64), (40, 33), (1000, 563)])
def test_farneback_flow_matches_the_oracle(ctx, W, H):
    """cv-decoder's arguments on region-motion content (flow discontinuities) and on a smooth camera rotation; sizes whose layers halve
    exactly, sizes with rounding in every layer, a size with two layers only, and one below the 32 px floor of the second layer."""
    fr = synth.luma_sequence(2, W, H, max_step=4, seed=W + H)
    f_o = oracle.farneback_flow(fr[0], fr[1])
    f_g, e_g = ctx.farneback_flow(fr[0], fr[1], want_entries=True)
    worst, same = _check(f_g, f_o)
    np.testing.assert_array_equal(e_g.view(np.uint32), oracle.flow_to_entries(f_g).view(np.uint32))
    print(f"{W}x{H}: max |d| {worst:.2e}, bit-identical flow components {same:.6f}")
    if min(W, H) >= 64:
        cl, _ = synth.rotation_clip([(0.1, 0.05, 0.2)], W, H, 60.0, seed=3)
        _check(ctx.farneback_flow(cl[0], cl[1]), oracle.farneback_flow(cl[0], cl[1]))


def test_farneback_parameters_other_than_cv_decoders(ctx):
    fr = synth.luma_sequence(2, 480, 270, max_step=3, seed=5)
    for kw in (dict(levels=3, winsize=9, iters=2, poly_n=5, poly_sigma=1.1), dict(levels=0, winsize=15, iters=1, poly_n=7, poly_sigma=1.5),
               dict(levels=5, winsize=13, iters=4, poly_n=3, poly_sigma=0.0)):
        # (the C ABI takes poly_sigma as a float: the oracle gets the same float32 value -- 1.1 is not representable, cv-decoder's 1.5 is)
        kw_o = dict(kw, poly_sigma=float(np.float32(kw["poly_sigma"])))
        worst, same = _check(ctx.farneback_flow(fr[0], fr[1], **kw), oracle.farneback_flow(fr[0], fr[1], **kw_o))
        print(f"{kw}: max |d| {worst:.2e} px, bit-identical flow components {same:.6f}")


def test_farneback_initial_flow(ctx):
    """OPTFLOW_USE_INITIAL_FLOW: cv-decoder hands its previous flow back in (cv-decoder/src/lib.rs:161-165)."""
    fr = synth.luma_sequence(3, 480, 270, max_step=3, seed=8)
    first = oracle.farneback_flow(fr[0], fr[1])
    _check(ctx.farneback_flow(fr[1], fr[2], init=first), oracle.farneback_flow(fr[1], fr[2], init=first))


def test_farneback_flat_and_identical_frames(ctx):
    flat = np.full((2, 128, 160), 77, np.uint8)
    assert np.array_equal(ctx.farneback_flow(flat[0], flat[1]), np.zeros((128, 160, 2), np.float32))       # singular system: det + 1e-3 -> zero flow
    fr = synth.luma_sequence(1, 320, 192, max_step=2, seed=1)
    f = ctx.farneback_flow(fr[0], fr[0])
    # identical frames: zero flow, except within a window of the last row / column, where the warp's "inside" test (x1 < w - 1)
    # drops the second image's linear terms even at zero displacement -- OpenCV's behaviour, restated as it is
    _check(f, oracle.farneback_flow(fr[0], fr[0]), tol=0.0)
    assert np.abs(f).max() < 0.1                                # (the coarse layers carry the border's effect far inward)


def test_farneback_rejects_what_it_has_no_kernel_for(ctx):
    from ofps_amd.runtime import OfpsHipError
    fr = synth.luma_sequence(2, 128, 96, max_step=2, seed=2)
    for kw in (dict(winsize=17), dict(winsize=12), dict(poly_n=16), dict(iters=0)):
        with pytest.raises(OfpsHipError):
            ctx.farneback_flow(fr[0], fr[1], **kw)
    _check(ctx.farneback_flow(fr[0], fr[1]), oracle.farneback_flow(fr[0], fr[1]))                            # the context stays usable


def test_farneback_1080p_pair_and_the_truth(ctx):
    """The size BASELINE quotes: against the oracle, and against the planted homography (median error of a few hundredths of a pixel)."""
    W, H = 1920, 1080
    cl, _ = synth.rotation_clip([(0.1, -0.05, 0.15)], W, H, 60.0, seed=4)
    f_o = oracle.farneback_flow(cl[0], cl[1])
    f_g = ctx.farneback_flow(cl[0], cl[1])
    worst, same = _check(f_g, f_o)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pos = np.stack([(xx.ravel() + .5) / W, (yy.ravel() + .5) / H], 1)
    truth = synth.rotation_delta(pos, W / H, 60.0, synth.euler_rot3(*np.radians((0.1, -0.05, 0.15))).T).reshape(H, W, 2) * [W, H]
    err = np.linalg.norm(f_g - truth, axis=2)
    print(f"1080p: max |d| vs oracle {worst:.2e}, identical bits {same:.6f}, median error vs the planted flow {np.median(err):.4f} px")
    assert np.median(err) < 0.05


def test_hip_flow_decoder_is_the_lk_decoders_output_stage_on_farnebacks_flow(ctx):
    """OFPS_HIP_FLOW_FARNEBACK in the decoder entry points: Farneback's flow -> per-pixel records -> cv-decoder's contrast mask ->
    densifier down-sampling to the capped grid, equal to the oracle's stages chained; pair call, stream form, read-ahead form, and
    the plugin mirror (hip_flow)."""
    from ofps_amd.plugins import HipFlowDecoder
    fr = synth.luma_sequence(4, 480, 270, max_step=3, seed=21)

    def chain(a, b, mask=True, fullres_records=False):
        flow = oracle.farneback_flow(a, b)
        ent = oracle.masked_flow_to_entries(flow, oracle.contrast_mask(b) if mask else None)
        return ent if fullres_records else oracle.densify_to_entries(ent, 150, 84)
    want = [chain(fr[k], fr[k + 1]) for k in range(3)]
    ent, grid = ctx.lk_decode(fr[0], fr[1], 5, 6, 3, contrast_mask=True, farneback=True)
    assert grid == (150, 84)
    np.testing.assert_array_equal(ent.view(np.uint32), want[0].view(np.uint32))
    ent, _ = ctx.lk_decode(fr[0], fr[1], 5, 6, 3, contrast_mask=False, fullres_records=True, farneback=True)
    np.testing.assert_array_equal(ent.view(np.uint32), chain(fr[0], fr[1], mask=False, fullres_records=True).view(np.uint32))
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[0], 5, 6, 3, contrast_mask=True, farneback=True) is None
    for k in range(1, 4):
        ent, _ = ctx.lk_push_frame(fr[k], 5, 6, 3, contrast_mask=True, farneback=True)
        np.testing.assert_array_equal(ent.view(np.uint32), want[k - 1].view(np.uint32))
    ctx.lk_reset()
    pins = [ctx.pinned_frame(270, 480) for _ in range(4)]
    for k in range(4):
        np.copyto(pins[k], fr[k])
    t = [ctx.lk_push_frame_async(pins[0], 5, 6, 3, contrast_mask=True, farneback=True), ctx.lk_push_frame_async(pins[1], 5, 6, 3, contrast_mask=True, farneback=True)]
    got = [ctx.lk_frame_wait(t[0])]
    t.append(ctx.lk_push_frame_async(pins[2], 5, 6, 3, contrast_mask=True, farneback=True))
    got.append(ctx.lk_frame_wait(t[1]))
    t.append(ctx.lk_push_frame_async(pins[3], 5, 6, 3, contrast_mask=True, farneback=True))
    got += [ctx.lk_frame_wait(t[2]), ctx.lk_frame_wait(t[3])]
    assert got[0] is None
    for k in range(1, 4):
        np.testing.assert_array_equal(got[k][0].view(np.uint32), want[k - 1].view(np.uint32))
    ctx.lk_reset()
    # the plugin mirror does what cv-decoder does (cv-decoder/src/lib.rs:161-165): from the second pair on, the previous flow is the initial flow
    dec = HipFlowDecoder(iter(fr))
    field = []
    assert dec.process_frame(field) is False
    flow = None
    for k in range(3):
        field = []
        assert dec.process_frame(field) is True
        flow = oracle.farneback_flow(fr[k], fr[k + 1], init=flow)
        e_o = oracle.densify_to_entries(oracle.masked_flow_to_entries(flow, oracle.contrast_mask(fr[k + 1])), 150, 84)
        np.testing.assert_array_equal(np.asarray(field, np.float32).view(np.uint32), e_o.view(np.uint32))
    dec.use_previous_flow = False                                   # "flags = 0" for every pair
    dec2 = HipFlowDecoder(iter(fr)); dec2.use_previous_flow = False
    assert dec2.process_frame([]) is False
    for k in range(3):
        field = []
        assert dec2.process_frame(field) is True
        np.testing.assert_array_equal(np.asarray(field, np.float32).view(np.uint32), want[k].view(np.uint32))
    dec.ctx.close(); dec2.ctx.close()


def test_hip_flow_stream_reuses_the_previous_frames_expansion(ctx):
    """Stream forms: the second frame's pyramid + polynomial expansion serve as the next pair's first (ofps_hip_flow_cache_hits counts it);
    records stay those of the pair call, whatever happens in between: a pair call through the same workspace, other parameters, a
    reset, a frame skipped by the caller (= a stream restart)."""
    fr = synth.luma_sequence(7, 352, 200, max_step=3, seed=33)

    def pair(a, b, levels=5, radius=6):
        return ctx.lk_decode(a, b, levels, radius, 3, contrast_mask=True, farneback=True)[0]
    want = [pair(fr[k], fr[k + 1]) for k in range(6)]
    grid = ctx.lk_decode(fr[0], fr[1], 5, 6, 3, contrast_mask=True, farneback=True)[1]
    np.testing.assert_array_equal(want[0].view(np.uint32),
                                  oracle.densify_to_entries(oracle.masked_flow_to_entries(oracle.farneback_flow(fr[0], fr[1]), oracle.contrast_mask(fr[1])), *grid).view(np.uint32))
    ctx.lk_reset()
    h0 = ctx.flow_cache_hits()
    assert ctx.lk_push_frame(fr[0], 5, 6, 3, contrast_mask=True, farneback=True) is None
    got = [ctx.lk_push_frame(fr[k], 5, 6, 3, contrast_mask=True, farneback=True)[0] for k in (1, 2, 3)]
    assert ctx.flow_cache_hits() - h0 == 3                            # every pair found its first frame's planes (round 6: a frame is expanded when it is pushed)
    for k in range(3):
        np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32))
    # a pair call in between uses two of the three plane slots (the oldest frames'): the stream's last frame keeps its own
    ctx.farneback_flow(fr[5], fr[6])
    e = ctx.lk_push_frame(fr[4], 5, 6, 3, contrast_mask=True, farneback=True)[0]
    assert ctx.flow_cache_hits() - h0 == 4
    np.testing.assert_array_equal(e.view(np.uint32), want[3].view(np.uint32))
    # other parameters (fewer layers: 352 x 200 has three at levels >= 2, two at levels = 1): recomputed
    e = ctx.lk_push_frame(fr[5], 1, 6, 3, contrast_mask=True, farneback=True)[0]
    assert ctx.flow_cache_hits() - h0 == 4
    np.testing.assert_array_equal(e.view(np.uint32), pair(fr[4], fr[5], levels=1).view(np.uint32))
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[2], 5, 6, 3, contrast_mask=True, farneback=True) is None
    e = ctx.lk_push_frame(fr[3], 5, 6, 3, contrast_mask=True, farneback=True)[0]
    np.testing.assert_array_equal(e.view(np.uint32), want[2].view(np.uint32))
    # read-ahead form: two tickets in flight
    pins = [ctx.pinned_frame(200, 352) for _ in range(3)]
    t = []
    for k in range(3):
        np.copyto(pins[k], fr[4 + k])
    h1 = ctx.flow_cache_hits()
    t.append(ctx.lk_push_frame_async(pins[0], 5, 6, 3, contrast_mask=True, farneback=True))     # continues the stream: pair (3, 4)
    t.append(ctx.lk_push_frame_async(pins[1], 5, 6, 3, contrast_mask=True, farneback=True))
    r0 = ctx.lk_frame_wait(t[0]); t.append(ctx.lk_push_frame_async(pins[2], 5, 6, 3, contrast_mask=True, farneback=True))
    r1 = ctx.lk_frame_wait(t[1]); r2 = ctx.lk_frame_wait(t[2])
    assert ctx.flow_cache_hits() - h1 == 3
    for r, k in ((r0, 3), (r1, 4), (r2, 5)):
        np.testing.assert_array_equal(r[0].view(np.uint32), want[k].view(np.uint32))
    ctx.lk_reset()


@pytest.mark.parametrize("W,H", [(3000, 80), (4400, 80), (2400, 270)])
def test_farneback_wide_frames_take_the_other_row_pitches(ctx, W, H):
    """The pyramid's row filter compiles three LDS row pitches (farneback.hip: kPyrRS0 / 1 / 2; the widest needs 135 KB of dynamic LDS):
    frames wider than 2,300 / 4,350 px go through the second / third, with one, one and three layers above the frame."""
    fr = synth.luma_sequence(2, W, H, max_step=3, seed=W)
    f_o = oracle.farneback_flow(fr[0], fr[1])
    worst, same = _check(ctx.farneback_flow(fr[0], fr[1]), f_o)
    print(f"{W}x{H}: max |d| {worst:.2e}, bit-identical flow components {same:.6f}")
    assert same == 1.0


def test_hip_flow_stream_with_the_previous_flow_as_initial_flow(ctx):
    """OFPS_HIP_FLOW_USE_PREVIOUS = OPTFLOW_USE_INITIAL_FLOW the way cv-decoder sets it (cv-decoder/src/lib.rs:161-165: its flow matrix
    persists, so every pair after the first starts from the previous pair's flow): stream form and read-ahead form against the oracle
    chained through `init`; a restart forgets the flow; a pair call in between does not disturb the stream; the flag without
    OFPS_HIP_FLOW_FARNEBACK is refused."""
    fr = synth.luma_sequence(6, 416, 240, max_step=3, seed=5)
    kw = dict(contrast_mask=True, farneback=True, use_previous=True)

    def chain(frames):
        flow, out = None, []
        for a, b in zip(frames[:-1], frames[1:]):
            flow = oracle.farneback_flow(a, b, init=flow)
            out.append(oracle.densify_to_entries(oracle.masked_flow_to_entries(flow, oracle.contrast_mask(b)), 150, 86))
        return out
    want = chain(fr)
    assert not np.array_equal(want[1], oracle.densify_to_entries(oracle.masked_flow_to_entries(oracle.farneback_flow(fr[1], fr[2]), oracle.contrast_mask(fr[2])), 150, 86))
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[0], 5, 6, 3, **kw) is None
    for k in range(1, 4):
        ent, grid = ctx.lk_push_frame(fr[k], 5, 6, 3, **kw)
        assert grid == (150, 86)
        np.testing.assert_array_equal(ent.view(np.uint32), want[k - 1].view(np.uint32))
        if k == 2:
            ctx.lk_decode(fr[4], fr[5], 5, 6, 3, contrast_mask=True, farneback=True)       # a pair on its own: zero initial flow, the stream's flow stays
    # restart: the first pair of the new stream starts from zero again
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[2], 5, 6, 3, **kw) is None
    np.testing.assert_array_equal(ctx.lk_push_frame(fr[3], 5, 6, 3, **kw)[0].view(np.uint32), chain(fr[2:4])[0].view(np.uint32))
    # read-ahead form, two tickets in flight
    ctx.lk_reset()
    pins = [ctx.pinned_frame(240, 416) for _ in range(6)]
    for k in range(6):
        np.copyto(pins[k], fr[k])
    t = [ctx.lk_push_frame_async(pins[0], 5, 6, 3, **kw), ctx.lk_push_frame_async(pins[1], 5, 6, 3, **kw)]
    got = []
    for k in range(2, 6):
        got.append(ctx.lk_frame_wait(t[k - 2]))
        t.append(ctx.lk_push_frame_async(pins[k], 5, 6, 3, **kw))
    got += [ctx.lk_frame_wait(t[4]), ctx.lk_frame_wait(t[5])]
    assert got[0] is None
    for k in range(1, 6):
        np.testing.assert_array_equal(got[k][0].view(np.uint32), want[k - 1].view(np.uint32))
    ctx.lk_reset()
    with pytest.raises(Exception):
        ctx.lk_push_frame(fr[0], 3, 4, 3, contrast_mask=True, use_previous=True)
    ctx.lk_reset()


@pytest.mark.parametrize("ahead", ["0", "1"])
def test_hip_flow_read_ahead_records_do_not_depend_on_where_the_new_frame_is_expanded(ctx, ahead):
    """OFPS_HIP_FB_PREPARE_AHEAD: a stream's new frame goes through the pyramid + expansion on the upload's stream when it is pushed (1, the
    default) or inside the pair's flow on the compute stream (0, round 5's order) -- an A/B switch, not a result switch: the oracle's chained
    records either way, and every frame after the first pair reuses the previous frame's expansion."""
    fr = synth.luma_sequence(6, 416, 240, max_step=3, seed=9)
    kw = dict(contrast_mask=True, farneback=True, use_previous=True)
    flow, want = None, []
    for a, b in zip(fr[:-1], fr[1:]):
        flow = oracle.farneback_flow(a, b, init=flow)
        want.append(oracle.densify_to_entries(oracle.masked_flow_to_entries(flow, oracle.contrast_mask(b)), 150, 86))
    ctx.set_option("OFPS_HIP_FB_PREPARE_AHEAD", ahead)
    try:
        ctx.lk_reset()
        pins = [ctx.pinned_frame(240, 416) for _ in range(6)]
        for k in range(6):
            np.copyto(pins[k], fr[k])
        h0 = ctx.flow_cache_hits()
        t = [ctx.lk_push_frame_async(pins[0], 5, 6, 3, **kw), ctx.lk_push_frame_async(pins[1], 5, 6, 3, **kw)]
        got = []
        for k in range(2, 6):
            got.append(ctx.lk_frame_wait(t[k - 2]))
            t.append(ctx.lk_push_frame_async(pins[k], 5, 6, 3, **kw))
        got += [ctx.lk_frame_wait(t[4]), ctx.lk_frame_wait(t[5])]
        assert got[0] is None
        for k in range(1, 6):
            np.testing.assert_array_equal(got[k][0].view(np.uint32), want[k - 1].view(np.uint32))
        assert ctx.flow_cache_hits() - h0 >= 4
    finally:
        ctx.set_option("OFPS_HIP_FB_PREPARE_AHEAD", None)
        ctx.lk_reset()
