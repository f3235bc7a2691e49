"""cv-decoder's frame front-end and its "Process Fullres" = false mode on the GPU (ofps_amd/csrc/frontend.hip, the decoder entry points of
lk.hip) against the oracle chain (oracle/frontend_oracle.c + the flows, mask and record loop), bit for bit, through the C ABI.
Reference: cv-decoder/src/lib.rs:98-135 (grid, resize, gray), :239-243,274-276 (one record per unmasked pixel of the reduced frame).
PARITY UNPINNED: the resize / colour arithmetic is OpenCV's, restated from its published code (OpenCV is not in the reference tree)."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle
from ofps_amd import mvec, synth
from ofps_amd.runtime import OfpsHipError

pytestmark = pytest.mark.gpu
FMT = {1: oracle.FMT_LUMA, 3: oracle.FMT_BGR}


@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


def _img(H, W, cn, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (H, W, cn), dtype=np.uint8)
    return a[:, :, 0].copy() if cn == 1 else a


def colour_clip(n, W, H, seed, max_step=3):
    """BGR frames whose three channels move together (so the gray frames have real motion) with per-channel texture"""
    y = synth.flatten_regions(synth.luma_sequence(n, W, H, max_step=max_step, seed=seed), region=max(24, W // 12), seed=seed + 1)
    rng = np.random.default_rng(seed)
    tint = rng.integers(-40, 41, (1, H, W, 3))
    return np.clip(y[..., None].astype(int) + tint, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("W,H,dw,dh,cn", [(1920, 1080, 150, 84, 3), (1920, 1080, 150, 84, 1), (3840, 2160, 150, 84, 3), (333, 77, 150, 34, 3),
                                           (97, 61, 40, 25, 4), (64, 48, 64, 48, 3), (300, 168, 150, 84, 3), (300, 168, 150, 84, 1), (8, 6, 3, 5, 1),
                                           (5, 7, 9, 11, 3), (2, 2, 1, 1, 1), (1, 1, 1, 1, 3), (640, 360, 150, 84, 4), (3, 1000, 1, 150, 1),
                                           (1921, 1081, 700, 394, 3)])
def test_resize_linear_is_bit_exact(ctx, W, H, dw, dh, cn):
    """general bilinear path (edge rule, clamped rows), exact 2 x 2 reductions, same size, enlargements, 1 / 3 / 4 channels"""
    a = _img(H, W, cn, W + 3 * H + cn)
    fmt = {1: ctx.FMT_LUMA, 3: ctx.FMT_BGR, 4: ctx.FMT_RGBA}[cn]
    np.testing.assert_array_equal(ctx.resize_linear(a, dw, dh, fmt), oracle.resize_linear(a, dw, dh))


@pytest.mark.parametrize("W,H", [(1920, 1080), (333, 77), (641, 359), (5, 3)])
def test_cv_frontend_all_formats(ctx, W, H):
    """[resize ->] gray for every pixel format, both modes: the reference's order (resize the colour frame, then convert)"""
    for fmt, cn in ((ctx.FMT_LUMA, 1), (ctx.FMT_BGR, 3), (ctx.FMT_RGBA, 4), (ctx.FMT_BGRA, 4)):
        a = _img(H, W, cn, W + H + fmt)
        for reduced in (False, True):
            g = ctx.cv_frontend(a, fmt, reduced)
            np.testing.assert_array_equal(g, oracle.cv_frontend(a, fmt, not reduced))
            assert g.shape == ((H, W) if not reduced else oracle.cv_grid(W, H)[::-1])
    assert ctx.cv_grid(W, H) == oracle.cv_grid(W, H)
    assert ctx.cv_grid(W, H, 2000, 150) == oracle.cv_grid(W, H, 2000, 150)


def test_cv_frontend_device_pointers_padded_stride(ctx):
    import torch
    W, H, pitch = 1001, 563, 3008
    a = _img(H, W, 3, 2)
    buf = np.zeros((H, pitch), np.uint8)
    buf[:, :3 * W] = a.reshape(H, 3 * W)
    d_in = torch.from_numpy(buf).cuda()
    d_out = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for reduced in (False, True):
        ow, oh = ctx.cv_frontend_dev(d_in.data_ptr(), W, H, pitch, ctx.FMT_BGR, reduced, 150, 150, d_out.data_ptr())
        ctx.sync()
        np.testing.assert_array_equal(d_out.cpu().numpy()[:ow * oh].reshape(oh, ow), oracle.cv_frontend(a, oracle.FMT_BGR, not reduced))
    # an unaligned base (the dword fast path must not be taken)
    d_in2 = torch.zeros(H * pitch + 1, dtype=torch.uint8, device="cuda")
    d_in2[1:] = d_in.reshape(-1)
    torch.cuda.synchronize()
    ctx.cv_frontend_dev(d_in2.data_ptr() + 1, W, H, pitch, ctx.FMT_BGR, False, 150, 150, d_out.data_ptr())
    ctx.sync()
    np.testing.assert_array_equal(d_out.cpu().numpy().reshape(H, W), oracle.to_gray(a))


@pytest.mark.parametrize("flow", ["farneback", "lk"])
@pytest.mark.parametrize("W,H,cn", [(1920, 1080, 3), (1920, 1080, 1), (641, 359, 3), (333, 177, 1)])
def test_reduced_decoder_matches_the_oracle_chain(ctx, flow, W, H, cn):
    """ "Process Fullres" = false: resize -> gray -> mask + flow on the reduced frames -> one record per unmasked reduced-frame pixel.
    1080p -> 150 x 84 and odd sizes; BGR and luma input; hip_flow (cv-decoder's own flow) and hip_lk."""
    clip = colour_clip(2, W, H, seed=W + cn)
    fr = clip if cn == 3 else clip[..., 1].copy()
    fmt = FMT[cn]
    far = flow == "farneback"
    args = (5, 6, 3) if far else (3, 4, 3)
    for mask in (True, False):
        rec_o, grid_o, _ = oracle.cv_decode(fr[0], fr[1], fmt, process_fullres=False, flow=flow, contrast_mask_on=mask)
        rec, grid = ctx.lk_decode(fr[0], fr[1], *args, contrast_mask=mask, farneback=far, reduced=True, fmt=fmt)
        assert grid == grid_o == oracle.cv_grid(W, H)
        np.testing.assert_array_equal(rec.view(np.uint32), rec_o.view(np.uint32))
        if not mask:
            assert len(rec) == grid[0] * grid[1]                 # 12,600 at 1080p: the reduced frame's pixels, not 2.07 M
    if (W, H) == (1920, 1080):
        assert grid == (150, 84)
    # the default mode with colour input: gray conversion at full resolution, then the densifier (bit-exact too)
    if cn == 3 and W <= 700:
        rec_o, grid_o, _ = oracle.cv_decode(fr[0], fr[1], fmt, process_fullres=True, flow=flow)
        rec, grid = ctx.lk_decode(fr[0], fr[1], *args, contrast_mask=True, farneback=far, fmt=fmt)
        assert grid == grid_o
        np.testing.assert_array_equal(rec.view(np.uint32), rec_o.view(np.uint32))


def test_reduced_stream_forms_and_mode_flip(ctx):
    """stream + read-ahead forms in the reduced mode equal the pair call; with OFPS_HIP_FLOW_USE_PREVIOUS the REDUCED flow is carried from pair
    to pair; flipping "Process Fullres" mid-stream restarts the stream (cv-decoder returns Ok(false): gray and old_gray differ in size,
    cv-decoder/src/lib.rs:156-158)."""
    W, H, F = 640, 360, 5
    fr = colour_clip(F, W, H, seed=8)
    kw = dict(contrast_mask=True, farneback=True, reduced=True, fmt=ctx.FMT_BGR)
    ctx.lk_reset()
    assert ctx.lk_push_frame(fr[0], 5, 6, 3, **kw) is None
    for k in range(1, F):
        rec, grid = ctx.lk_push_frame(fr[k], 5, 6, 3, **kw)
        rec_o, grid_o = ctx.lk_decode(fr[k - 1], fr[k], 5, 6, 3, **kw)
        assert grid == grid_o == (150, 84)
        np.testing.assert_array_equal(rec.view(np.uint32), rec_o.view(np.uint32))
    # read-ahead, two tickets in flight, page-locked colour frames
    pins = [ctx.pinned_frame(H, 3 * W).reshape(H, W, 3) for _ in range(3)]
    ctx.lk_reset()
    got, tickets = [], []
    for k in range(F):
        np.copyto(pins[k % 3], fr[k])
        tickets.append(ctx.lk_push_frame_async(pins[k % 3], 5, 6, 3, **kw))
        if k >= 1:
            got.append(ctx.lk_frame_wait(tickets[k - 1]))
    got.append(ctx.lk_frame_wait(tickets[-1]))
    assert got[0] is None
    for k in range(1, F):
        rec_o, _ = ctx.lk_decode(fr[k - 1], fr[k], 5, 6, 3, **kw)
        np.testing.assert_array_equal(got[k][0].view(np.uint32), rec_o.view(np.uint32))
    # the previous REDUCED flow as the initial flow, as cv-decoder's self.flow (150 x 84 in this mode)
    ctx.lk_reset()
    kwp = dict(kw, use_previous=True)
    assert ctx.lk_push_frame(fr[0], 5, 6, 3, **kwp) is None
    flow = None
    for k in range(1, F):
        rec, _ = ctx.lk_push_frame(fr[k], 5, 6, 3, **kwp)
        rec_o, _, flow = oracle.cv_decode(fr[k - 1], fr[k], oracle.FMT_BGR, process_fullres=False, init=flow)
        np.testing.assert_array_equal(rec.view(np.uint32), rec_o.view(np.uint32))
    # mode flip: the next frame starts a new stream
    assert ctx.lk_push_frame(fr[0], 5, 6, 3, **dict(kw, reduced=False)) is None
    assert ctx.lk_push_frame(fr[1], 5, 6, 3, **dict(kw, reduced=False)) is not None
    assert ctx.lk_push_frame(fr[2], 5, 6, 3, **kw) is None
    assert ctx.lk_push_frame(fr[3][..., 1].copy(), 5, 6, 3, **dict(kw, fmt=ctx.FMT_LUMA)) is None      # another pixel format: a new stream too
    ctx.lk_reset()
    # contradictory flags and a stride too small for the format are refused
    with pytest.raises(OfpsHipError):
        ctx.lk_decode(fr[0], fr[1], 5, 6, 3, reduced=True, fullres_records=True, fmt=ctx.FMT_BGR)
    with pytest.raises(ValueError):
        ctx.lk_decode(fr[0], fr[1], 5, 6, 3, reduced=True)                  # a colour array announced as luma


def test_plugin_mirror_flips_process_fullres(ctx):
    """HipFlowDecoder with cv-decoder's property: false -> ~12.6 k records per frame from the reduced frames, true -> the densifier's cells;
    the frame after the flip has no vectors; skipped frames keep the previous flow as the initial flow (cv-decoder/src/lib.rs:161-165)."""
    from ofps_amd.plugins import HipFlowDecoder
    W, H, F = 640, 360, 7
    fr = colour_clip(F, W, H, seed=15)
    dec = HipFlowDecoder(iter(fr), frame_format=ctx.FMT_BGR)
    assert ("Process Fullres", "bool", True, None, None) in dec.props()
    assert dec.set_prop("Process Fullres", False)
    field = []
    assert dec.process_frame(field) is False
    flow = None
    for k in (1, 2):
        field = []
        assert dec.process_frame(field) is True
        rec_o, _, flow = oracle.cv_decode(fr[k - 1], fr[k], oracle.FMT_BGR, process_fullres=False, init=flow)
        assert 0 < len(rec_o) <= 12600
        np.testing.assert_array_equal(np.asarray(field, np.float32).view(np.uint32), rec_o.view(np.uint32))
    assert dec.get_aspect() == (150, 84)                        # the reduced gray frame's size (cv-decoder/src/lib.rs:296-298)
    shown = []
    field = []
    dec2 = HipFlowDecoder(iter(fr[:2]), frame_format=ctx.FMT_BGR); dec2.set_prop("Process Fullres", False)
    dec2.process_frame(field, shown)                            # the frame handed out is cv-decoder's `self.frame`: the RESIZED colour frame (:144-154)
    np.testing.assert_array_equal(shown[0], oracle.resize_linear(fr[0], 150, 84))
    dec.set_prop("Process Fullres", True)                       # frames 2, 3 are the pair now, at full resolution, from zero flow (a new stream)
    field = []
    assert dec.process_frame(field) is True
    rec_o, _, flow = oracle.cv_decode(fr[2], fr[3], oracle.FMT_BGR, process_fullres=True)
    np.testing.assert_array_equal(np.asarray(field, np.float32).view(np.uint32), rec_o.view(np.uint32))
    assert dec.get_aspect() == (640, 360)
    field = []
    assert dec.process_frame(field, skip_frames=1) is True      # frames 4 read and dropped, pair (4, 5): the flow of (2, 3) is still the initial flow
    rec_o, _, flow = oracle.cv_decode(fr[4], fr[5], oracle.FMT_BGR, process_fullres=True, init=flow)
    np.testing.assert_array_equal(np.asarray(field, np.float32).view(np.uint32), rec_o.view(np.uint32))


def test_cpp_host_flips_the_property(tmp_path):
    """create_decoder("hip_flow", "...&fmt=bgr") through the C++ host layer; "Process Fullres=false@3" flips the property before frame 3"""
    from ofps_amd import build as hip_build
    tool = os.path.join(os.path.dirname(hip_build.LIB), "host", "ofps_hip_tool")
    W, H, F = 320, 180, 6
    fr = colour_clip(F, W, H, seed=23)
    raw = tmp_path / "clip.bgr"
    raw.write_bytes(fr.tobytes())
    out = tmp_path / "clip.mvec"
    info = json.loads(subprocess.run([tool, "extract", "hip_flow", f"{raw}?w={W}&h={H}&fmt=bgr", str(out), "100", "Process Fullres=false@3"],
                                     check=True, capture_output=True, text=True).stdout)
    frames = list(mvec.read_frames(open(out, "rb")))
    assert info["frames"] == F and len(frames[0]) == 0
    flow = None
    for k in (1, 2):
        rec_o, _, flow = oracle.cv_decode(fr[k - 1], fr[k], oracle.FMT_BGR, process_fullres=True, init=flow)
        np.testing.assert_array_equal(frames[k].view(np.uint32), rec_o.view(np.uint32))
    # frame 3: the property changed -> the host re-pushes frame 2 in the new mode (a new stream: zero initial flow) and frame 3 completes the pair
    flow = None
    gw, gh = oracle.cv_grid(W, H)
    for k in (3, 4, 5):
        rec_o, _, flow = oracle.cv_decode(fr[k - 1], fr[k], oracle.FMT_BGR, process_fullres=False, init=flow)
        assert len(rec_o) <= gw * gh
        np.testing.assert_array_equal(frames[k].view(np.uint32), rec_o.view(np.uint32))


def test_farneback_limits_are_refused_with_the_first_frame(ctx):
    """a stream must not accept its first frame and fail every later one; 4K with six layers above the frame runs (159-tap blur)."""
    fr = synth.luma_sequence(2, 320, 180, max_step=2, seed=3)
    ctx.lk_reset()
    with pytest.raises(OfpsHipError):
        ctx.lk_push_frame(fr[0], 5, 8, 3, farneback=True)                   # winsize 17 > 15: refused with the FIRST frame
    with pytest.raises(OfpsHipError):
        ctx.lk_decode(fr[0], fr[1], 5, 8, 3, farneback=True)
    assert ctx.lk_push_frame(fr[0], 5, 7, 3, farneback=True) is None         # winsize 15 is the largest
    assert ctx.lk_push_frame(fr[1], 5, 7, 3, farneback=True) is not None
    ctx.lk_reset()
    big = synth.luma_sequence(2, 3840, 2160, max_step=6, seed=5)
    f_g = ctx.farneback_flow(big[0], big[1], levels=6)
    f_o = oracle.farneback_flow(big[0], big[1], levels=6)
    assert len(oracle.farneback_layers(3840, 2160, 6)) == 7
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))


def test_decoders_take_padded_strides_and_every_format(ctx):
    """rows `stride` bytes apart with stride > W * channels and not a multiple of 4 (the dword fast paths must not be taken), RGBA / BGRA
    frames, a frame smaller than the cap in the reduced mode (the "resize" is a copy, the records are still per pixel, cv-decoder/src/lib.rs:
    98-121,274-276) -- through the C ABI directly"""
    import ctypes as C
    from ofps_amd import _lib
    lib = _lib.load()
    W, H = 333, 177
    clip = colour_clip(2, W, H, seed=31)
    rec_o, grid_o, _ = oracle.cv_decode(clip[0], clip[1], oracle.FMT_BGR, process_fullres=False)
    stride = 3 * W + 13
    buf = np.zeros((2, H, stride), np.uint8)
    buf[:, :, :3 * W] = clip.reshape(2, H, 3 * W)
    out = np.zeros((150 * 150, 4), np.float32)
    n = C.c_size_t(0); gw = C.c_int(0); gh = C.c_int(0)
    u8 = C.POINTER(C.c_uint8)
    flags = ctx.LK_CONTRAST_MASK | ctx.FLOW_FARNEBACK | ctx.LK_REDUCED | (ctx.FMT_BGR << 8)
    rc = lib.ofps_hip_lk_decode(ctx._h, buf[0].ctypes.data_as(u8), buf[1].ctypes.data_as(u8), W, H, stride, 5, 6, 3, 150, 150, flags,
                                out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n), C.byref(gw), C.byref(gh))
    assert rc == 0 and (gw.value, gh.value) == grid_o
    np.testing.assert_array_equal(out[:n.value].view(np.uint32), rec_o.view(np.uint32))
    # a stride that cannot hold a row is refused
    rc = lib.ofps_hip_lk_decode(ctx._h, buf[0].ctypes.data_as(u8), buf[1].ctypes.data_as(u8), W, H, 3 * W - 1, 5, 6, 3, 150, 150, flags,
                                out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n), C.byref(gw), C.byref(gh))
    assert rc == -1
    # RGBA / BGRA: the same gray frames as the BGR clip, hence the same records
    rgba = np.concatenate([clip[..., ::-1], np.full(clip.shape[:3] + (1,), 255, np.uint8)], 3)
    bgra = np.concatenate([clip, np.full(clip.shape[:3] + (1,), 3, np.uint8)], 3)
    for fr, fmt in ((rgba, ctx.FMT_RGBA), (bgra, ctx.FMT_BGRA)):
        rec, grid = ctx.lk_decode(fr[0], fr[1], 5, 6, 3, contrast_mask=True, farneback=True, reduced=True, fmt=fmt)
        assert grid == grid_o
        np.testing.assert_array_equal(rec.view(np.uint32), rec_o.view(np.uint32))
        rec_f, _ = ctx.lk_decode(fr[0], fr[1], 5, 6, 3, contrast_mask=True, farneback=True, fmt=fmt)
        rec_fo, _, _ = oracle.cv_decode(clip[0], clip[1], oracle.FMT_BGR, process_fullres=True)
        np.testing.assert_array_equal(rec_f.view(np.uint32), rec_fo.view(np.uint32))
    # a frame under the cap: grid = frame, one record per unmasked pixel, no densifier
    small = colour_clip(2, 96, 64, seed=5)
    rec, grid = ctx.lk_decode(small[0], small[1], 5, 6, 3, contrast_mask=False, farneback=True, reduced=True, fmt=ctx.FMT_BGR)
    rec_so, grid_so, _ = oracle.cv_decode(small[0], small[1], oracle.FMT_BGR, process_fullres=False, contrast_mask_on=False)
    assert grid == grid_so == (96, 64) and len(rec) == 96 * 64
    np.testing.assert_array_equal(rec.view(np.uint32), rec_so.view(np.uint32))


def test_hip_flow_parameter_change_with_a_ticket_in_flight_is_refused(ctx):
    """a new frame is expanded ahead of its pair into the plane slot of the oldest frame; other Farneback parameters re-plan that workspace,
    which a flow still in flight would be reading: refused like a geometry change, accepted once the ticket is collected"""
    fr = synth.luma_sequence(4, 320, 180, max_step=2, seed=3)
    pins = [ctx.pinned_frame(180, 320) for _ in range(4)]
    for k in range(4):
        np.copyto(pins[k], fr[k])
    kw = dict(contrast_mask=True, farneback=True)
    ctx.lk_reset()
    t0 = ctx.lk_push_frame_async(pins[0], 5, 6, 3, **kw)
    t1 = ctx.lk_push_frame_async(pins[1], 5, 6, 3, **kw)
    assert ctx.lk_frame_wait(t0) is None
    with pytest.raises(OfpsHipError) as ei:
        ctx.lk_push_frame_async(pins[2], 2, 4, 3, **kw)                       # t1 is in flight
    assert "in flight" in str(ei.value)
    got = ctx.lk_frame_wait(t1)
    want, _ = ctx.lk_decode(fr[0], fr[1], 5, 6, 3, **kw)
    np.testing.assert_array_equal(got[0].view(np.uint32), want.view(np.uint32))
    t2 = ctx.lk_push_frame_async(pins[2], 2, 4, 3, **kw)                      # nothing in flight: accepted, the pair (1, 2) with the new parameters
    got = ctx.lk_frame_wait(t2)
    want, _ = ctx.lk_decode(fr[1], fr[2], 2, 4, 3, **kw)
    np.testing.assert_array_equal(got[0].view(np.uint32), want.view(np.uint32))
    ctx.lk_reset()
