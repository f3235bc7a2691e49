"""Thin Python front-end over the C ABI (include/ofps_hip.h): context handling plus numpy and
device-pointer call wrappers.  All compute happens in libofps_hip.so; nothing here has a CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import OfpsHipError


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class HipContext:
    """One ofps_hip_ctx (one per plugin instance, ofps/src/plugins/mod.rs:244-278 'Send, not Sync')."""

    def __init__(self, device: int = 0, test_hooks: bool = False):
        # test_hooks: bind libofps_hip_testhooks.so (fault injectors compiled in) instead of the product library
        self._lib = _lib.load_test_hooks() if test_hooks else _lib.load()
        h = C.c_void_p()
        rc = self._lib.ofps_hip_init(device, C.byref(h))
        if rc != 0:
            raise OfpsHipError(rc, (self._lib.ofps_hip_last_error(None) or b"").decode())
        self._h = h
        self.device = device
        self._pinned = []

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self._lib.ofps_hip_host_free(self._h, C.c_void_p(p))
            self._pinned = []
            self._lib.ofps_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise OfpsHipError(rc, (self._lib.ofps_hip_last_error(self._h) or b"").decode())

    # ---- plumbing
    def set_stream(self, stream_ptr: int):
        """Enqueue on a caller-owned hipStream_t; 0 is HIP's default stream (torch's default)."""
        self._check(self._lib.ofps_hip_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_own_stream(self):
        self._check(self._lib.ofps_hip_use_own_stream(self._h))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def sync(self):
        self._check(self._lib.ofps_hip_sync(self._h))

    def set_option(self, name: str, value=None):
        """Diagnostic / A-B switch by its environment-variable name; None restores the default."""
        v = None if value is None else str(value).encode()
        self._check(self._lib.ofps_hip_set_option(self._h, name.encode(), v))

    def almeida_recoveries(self) -> int:
        n = C.c_uint64(0)
        self._check(self._lib.ofps_hip_almeida_recoveries(self._h, C.byref(n)))
        return int(n.value)

    # device memory for hosts without a HIP binding of their own (what the Rust shim uses for resident chains)
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p(0)
        self._check(self._lib.ofps_hip_malloc(self._h, nbytes, C.byref(p)))
        return int(p.value)

    def free(self, dptr: int):
        self._check(self._lib.ofps_hip_free(self._h, C.c_void_p(dptr)))

    def memcpy_h2d(self, dptr: int, host: np.ndarray):
        a = np.ascontiguousarray(host)
        self._check(self._lib.ofps_hip_memcpy_h2d(self._h, C.c_void_p(dptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def memcpy_d2h(self, host: np.ndarray, dptr: int):
        assert host.flags["C_CONTIGUOUS"]
        self._check(self._lib.ofps_hip_memcpy_d2h(self._h, host.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), host.nbytes))

    def get_stream(self) -> int:
        return int(self._lib.ofps_hip_get_stream(self._h) or 0)

    def timer_start(self):
        self._check(self._lib.ofps_hip_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        self._check(self._lib.ofps_hip_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    # ---- N1
    SAD_EXHAUSTIVE, SAD_PRUNED = 0, 1

    def set_sad_mode(self, mode: int):
        """0 = exhaustive (default), 1 = exact search with partial-distortion elimination (content-dependent run time)."""
        self._check(self._lib.ofps_hip_set_sad_mode(self._h, mode))

    def sad_pruned_overflow_strips(self) -> int:
        n = C.c_uint32(0)
        self._check(self._lib.ofps_hip_sad_pruned_overflow_strips(self._h, C.byref(n)))
        return int(n.value)

    def sad_flow(self, prev: np.ndarray, cur: np.ndarray, block: int, search_range: int, want_best=False):
        prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
        assert prev.shape == cur.shape and prev.ndim == 2
        H, W = prev.shape
        nb = int(self._lib.ofps_hip_sad_block_count(W, H, block))
        ent = np.zeros((max(nb, 1), 4), np.float32)
        best = np.zeros((max(nb, 1), 3), np.int32)
        n_out = C.c_size_t(0)
        u8 = C.POINTER(C.c_uint8)
        self._check(self._lib.ofps_hip_sad_flow(self._h, prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W,
                                                block, search_range, _fp(ent),
                                                best.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_out)))
        assert n_out.value == nb
        return (ent[:nb], best[:nb]) if want_best else ent[:nb]

    def sad_flow_dev(self, d_frames: int, n_frames: int, W: int, H: int, stride: int, frame_pitch: int, ref_mode: int,
                     block: int, search_range: int, d_out_entries: int, d_out_best: int | None = None):
        self._check(self._lib.ofps_hip_sad_flow_dev(self._h, C.c_void_p(d_frames), n_frames, W, H, stride, frame_pitch,
                                                    ref_mode, block, search_range, C.c_void_p(d_out_entries),
                                                    C.c_void_p(d_out_best or 0)))

    # ---- N2
    def lk_flow(self, prev: np.ndarray, cur: np.ndarray, levels=3, radius=4, iters=3, want_entries=False):
        prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
        H, W = prev.shape
        flow = np.zeros((H, W, 2), np.float32)
        ent = np.zeros((H * W, 4), np.float32) if want_entries else None
        u8 = C.POINTER(C.c_uint8)
        self._check(self._lib.ofps_hip_lk_flow(self._h, prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W, levels, radius,
                                               iters, _fp(flow), _fp(ent) if want_entries else None))
        return (flow, ent) if want_entries else flow

    def farneback_flow(self, prev: np.ndarray, cur: np.ndarray, levels=5, winsize=13, iters=3, poly_n=7, poly_sigma=1.5, init=None,
                       want_entries=False):
        """Farneback's dense flow with cv-decoder's arguments as defaults (cv-decoder/src/lib.rs:188-199) -> flow[H, W, 2]
        (dx, dy): prev(x, y) ~ cur(x + dx, y + dy) [, records[H*W, 4]].  init: None or a [H, W, 2] flow to start from."""
        prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
        H, W = prev.shape
        flow = np.zeros((H, W, 2), np.float32)
        ent = np.zeros((H * W, 4), np.float32) if want_entries else None
        ini = None if init is None else np.ascontiguousarray(init, np.float32).reshape(H, W, 2)
        u8 = C.POINTER(C.c_uint8)
        self._check(self._lib.ofps_hip_farneback_flow(self._h, prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W, levels, winsize, iters,
                                                      poly_n, poly_sigma, _fp(ini) if ini is not None else None, _fp(flow),
                                                      _fp(ent) if want_entries else None))
        return (flow, ent) if want_entries else flow

    def farneback_flow_dev(self, d_prev: int, d_cur: int, W: int, H: int, stride: int, levels=5, winsize=13, iters=3, poly_n=7,
                           poly_sigma=1.5, d_init: int | None = None, d_out_flow: int | None = None, d_out_entries: int | None = None):
        self._check(self._lib.ofps_hip_farneback_flow_dev(self._h, C.c_void_p(d_prev), C.c_void_p(d_cur), W, H, stride, levels, winsize, iters,
                                                          poly_n, poly_sigma, C.c_void_p(d_init or 0), C.c_void_p(d_out_flow or 0),
                                                          C.c_void_p(d_out_entries or 0)))

    def lk_spec_revision(self) -> int:
        return int(self._lib.ofps_hip_lk_spec_revision())

    def lk_helped_tiles(self) -> int:
        """Tiles of the one-launch pyramid flow that a waiting child computed itself on this context (0 on an in-order dispatcher; costs
        time, never bits; diagnostics, synchronises)."""
        n = C.c_uint64(0)
        self._check(self._lib.ofps_hip_lk_helped_tiles(self._h, C.byref(n)))
        return int(n.value)

    def flow_cache_hits(self) -> int:
        """hip_flow stream forms: calls of this context that found their first frame's pyramid + expansion already on the device."""
        n = C.c_uint64(0)
        self._check(self._lib.ofps_hip_flow_cache_hits(self._h, C.byref(n)))
        return int(n.value)

    def lk_flow_init(self, prev: np.ndarray, cur: np.ndarray, levels, radius, iters, init: np.ndarray):
        """ofps_hip_lk_flow_init_dev through library-owned device buffers: `init` [h_L, w_L, 2] is the coarsest level's
        starting flow.  -> flow[H, W, 2]."""
        prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
        init = np.ascontiguousarray(init, np.float32)
        H, W = prev.shape
        flow = np.zeros((H, W, 2), np.float32)
        bufs = [self.malloc(prev.nbytes), self.malloc(cur.nbytes), self.malloc(init.nbytes), self.malloc(flow.nbytes)]
        try:
            self.memcpy_h2d(bufs[0], prev); self.memcpy_h2d(bufs[1], cur); self.memcpy_h2d(bufs[2], init)
            self._check(self._lib.ofps_hip_lk_flow_init_dev(self._h, C.c_void_p(bufs[0]), C.c_void_p(bufs[1]), W, H, W, levels, radius,
                                                            iters, C.c_void_p(bufs[2]), C.c_void_p(bufs[3]), C.c_void_p(0)))
            self.memcpy_d2h(flow, bufs[3])
        finally:
            for b in bufs:
                self.free(b)
        return flow

    LK_CONTRAST_MASK, LK_FULLRES_RECORDS, FLOW_FARNEBACK, FLOW_USE_PREVIOUS, LK_REDUCED = 1, 2, 4, 8, 16
    FMT_LUMA, FMT_BGR, FMT_RGBA, FMT_BGRA = 0, 1, 2, 3

    def _lk_flags(self, contrast_mask, fullres_records, farneback, use_previous, reduced, fmt) -> int:
        return ((self.LK_CONTRAST_MASK if contrast_mask else 0) | (self.LK_FULLRES_RECORDS if fullres_records else 0) |
                (self.FLOW_FARNEBACK if farneback else 0) | (self.FLOW_USE_PREVIOUS if use_previous else 0) |
                (self.LK_REDUCED if reduced else 0) | (int(fmt) << 8))

    def _frame_geometry(self, frame: np.ndarray, fmt: int):
        """-> (W, H, row pitch in bytes) of a C-contiguous u8 frame [H, W] (luma) or [H, W, channels]"""
        cn = int(self._lib.ofps_hip_frame_channels(int(fmt)))
        if cn == 0:
            raise ValueError(f"unknown frame format {fmt}")
        if (frame.ndim == 2 and cn != 1) or (frame.ndim == 3 and frame.shape[2] != cn) or frame.ndim not in (2, 3):
            raise ValueError(f"frame of shape {frame.shape} is not a format-{fmt} frame ({cn} bytes per pixel)")
        H, W = frame.shape[:2]
        return W, H, W * cn

    @staticmethod
    def _lk_capacity(W, H, max_w, max_h, fullres_records) -> int:
        return W * H if fullres_records else min(max_w, W) * min(max_h, H)

    def cv_grid(self, W: int, H: int, max_w: int = 150, max_h: int = 150):
        """cv-decoder/src/lib.rs:98-121: the capped record grid (and the size "Process Fullres" = false resizes the frames to)."""
        gw = C.c_int(0); gh = C.c_int(0)
        rc = self._lib.ofps_hip_cv_grid(W, H, max_w, max_h, C.byref(gw), C.byref(gh))
        if rc != 0:
            raise ValueError("ofps_hip_cv_grid: bad arguments")
        return gw.value, gh.value

    def resize_linear(self, img: np.ndarray, dw: int, dh: int, fmt: int = 0) -> np.ndarray:
        """imgproc::resize(.., INTER_LINEAR) of an 8-bit frame, channels kept (cv-decoder/src/lib.rs:124-133)."""
        img = np.ascontiguousarray(img, np.uint8)
        W, H, pitch = self._frame_geometry(img, fmt)
        out = np.zeros((dh, dw) + img.shape[2:], np.uint8)
        u8 = C.POINTER(C.c_uint8)
        self._check(self._lib.ofps_hip_resize_linear(self._h, img.ctypes.data_as(u8), W, H, pitch, int(fmt), out.ctypes.data_as(u8), dw, dh))
        return out

    def cv_frontend(self, frame: np.ndarray, fmt: int = 0, reduced: bool = False, max_w: int = 150, max_h: int = 150) -> np.ndarray:
        """[resize to the capped grid ->] gray: what cv-decoder leaves in `self.gray` for one frame (cv-decoder/src/lib.rs:98-135)."""
        frame = np.ascontiguousarray(frame, np.uint8)
        W, H, pitch = self._frame_geometry(frame, fmt)
        out = np.zeros(W * H, np.uint8)
        ow = C.c_int(0); oh = C.c_int(0)
        u8 = C.POINTER(C.c_uint8)
        self._check(self._lib.ofps_hip_cv_frontend(self._h, frame.ctypes.data_as(u8), W, H, pitch, int(fmt), int(bool(reduced)), max_w, max_h,
                                                   out.ctypes.data_as(u8), C.byref(ow), C.byref(oh)))
        return out[:ow.value * oh.value].reshape(oh.value, ow.value).copy()

    def cv_frontend_dev(self, d_frame: int, W: int, H: int, stride: int, fmt: int, reduced: bool, max_w: int, max_h: int, d_out_gray: int):
        ow = C.c_int(0); oh = C.c_int(0)
        self._check(self._lib.ofps_hip_cv_frontend_dev(self._h, C.c_void_p(d_frame), W, H, stride, int(fmt), int(bool(reduced)), max_w, max_h,
                                                       C.c_void_p(d_out_gray), C.byref(ow), C.byref(oh)))
        return ow.value, oh.value

    def lk_decode(self, prev: np.ndarray, cur: np.ndarray, levels=3, radius=4, iters=3, max_w=150, max_h=150, farneback=False, use_previous=False,
                  contrast_mask=False, fullres_records=False, reduced=False, fmt=0):
        """-> (entries[n,4], (grid_w, grid_h)): what a hip_lk Decoder appends per frame (cv-decoder/src/lib.rs:82-294).
        reduced: cv-decoder's "Process Fullres" = false; fmt: the frames' pixel format (FMT_*; colour frames are [H, W, channels])."""
        prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
        assert prev.shape == cur.shape
        W, H, pitch = self._frame_geometry(prev, fmt)
        out = np.zeros((self._lk_capacity(W, H, max_w, max_h, fullres_records), 4), np.float32)
        n = C.c_size_t(0); gw = C.c_int(0); gh = C.c_int(0)
        u8 = C.POINTER(C.c_uint8)
        flags = self._lk_flags(contrast_mask, fullres_records, farneback, use_previous, reduced, fmt)
        self._check(self._lib.ofps_hip_lk_decode(self._h, prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, pitch, levels, radius,
                                                 iters, max_w, max_h, flags, _fp(out), C.byref(n), C.byref(gw), C.byref(gh)))
        return out[:n.value].copy(), (gw.value, gh.value)

    def lk_push_frame(self, frame: np.ndarray, levels=3, radius=4, iters=3, max_w=150, max_h=150, contrast_mask=False,
                      fullres_records=False, farneback=False, use_previous=False, reduced=False, fmt=0):
        """Stream form of lk_decode: the frame is uploaded once and is the next call's previous frame.
        -> None for the first frame of a stream, else (entries[n,4], (grid_w, grid_h))."""
        frame = np.ascontiguousarray(frame, np.uint8)
        W, H, pitch = self._frame_geometry(frame, fmt)
        out = np.zeros((self._lk_capacity(W, H, max_w, max_h, fullres_records), 4), np.float32)
        n = C.c_size_t(0); gw = C.c_int(0); gh = C.c_int(0); have = C.c_int(0)
        flags = self._lk_flags(contrast_mask, fullres_records, farneback, use_previous, reduced, fmt)
        self._check(self._lib.ofps_hip_lk_push_frame(self._h, frame.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, pitch, levels, radius,
                                                     iters, max_w, max_h, flags, _fp(out), C.byref(n), C.byref(gw), C.byref(gh),
                                                     C.byref(have)))
        if not have.value:
            return None
        return out[:n.value].copy(), (gw.value, gh.value)

    def lk_push_frame_async(self, frame: np.ndarray, levels=3, radius=4, iters=3, max_w=150, max_h=150, contrast_mask=False,
                            fullres_records=False, farneback=False, use_previous=False, reduced=False, fmt=0) -> int:
        """Read-ahead form: returns a ticket once the upload, the flow and the output stage are enqueued.  `frame` must be
        C-contiguous u8 and stay alive (ideally page-locked: pinned_frame) until lk_frame_wait(ticket)."""
        assert frame.dtype == np.uint8 and frame.flags["C_CONTIGUOUS"]
        W, H, pitch = self._frame_geometry(frame, fmt)
        flags = self._lk_flags(contrast_mask, fullres_records, farneback, use_previous, reduced, fmt)
        t = C.c_int(0)
        self._check(self._lib.ofps_hip_lk_push_frame_async(self._h, frame.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, pitch, levels, radius,
                                                           iters, max_w, max_h, flags, C.byref(t)))
        self._lk_cap = getattr(self, "_lk_cap", {})
        self._lk_cap[t.value] = self._lk_capacity(W, H, max_w, max_h, fullres_records)
        return t.value

    def lk_frame_wait(self, ticket: int, out: np.ndarray | None = None):
        """-> None for the first frame of a stream, else (entries[n,4], (grid_w, grid_h)).  `out`: a caller buffer of the
        capacity lk_decode documents (the result is then a view of it)."""
        cap = getattr(self, "_lk_cap", {}).pop(ticket, 1)    # an unknown ticket: the library says so
        if out is None:
            out = np.zeros((cap, 4), np.float32)
        assert out.dtype == np.float32 and out.size >= 4 * cap and out.flags["C_CONTIGUOUS"]
        n = C.c_size_t(0); gw = C.c_int(0); gh = C.c_int(0); have = C.c_int(0)
        self._check(self._lib.ofps_hip_lk_frame_wait(self._h, ticket, _fp(out), C.byref(n), C.byref(gw), C.byref(gh), C.byref(have)))
        if not have.value:
            return None
        return out.reshape(-1, 4)[:n.value], (gw.value, gh.value)

    def lk_reset(self):
        self._check(self._lib.ofps_hip_lk_reset(self._h))

    def lk_rewind(self):
        """forget the stream's frames, keep its last flow as the next pair's initial flow (a decoder that skipped frames)"""
        self._check(self._lib.ofps_hip_lk_rewind(self._h))

    def contrast_mask(self, gray: np.ndarray) -> np.ndarray:
        """cv-decoder's Sobel/threshold/dilate mask (cv-decoder/src/lib.rs:203-237) -> u8[H, W], 1 = keep."""
        g = np.ascontiguousarray(gray, np.uint8)
        H, W = g.shape
        out = np.zeros((H, W), np.uint8)
        u8 = C.POINTER(C.c_uint8)
        self._check(self._lib.ofps_hip_contrast_mask(self._h, g.ctypes.data_as(u8), W, H, W, out.ctypes.data_as(u8)))
        return out

    def contrast_mask_dev(self, d_gray: int, W: int, H: int, stride: int, d_out_mask: int):
        self._check(self._lib.ofps_hip_contrast_mask_dev(self._h, C.c_void_p(d_gray), W, H, stride, C.c_void_p(d_out_mask)))

    def lk_flow_dev(self, d_prev: int, d_cur: int, W: int, H: int, stride: int, levels: int, radius: int, iters: int,
                    d_out_flow: int | None, d_out_entries: int | None):
        self._check(self._lib.ofps_hip_lk_flow_dev(self._h, C.c_void_p(d_prev), C.c_void_p(d_cur), W, H, stride, levels, radius,
                                                   iters, C.c_void_p(d_out_flow or 0), C.c_void_p(d_out_entries or 0)))

    # ---- A1-A4
    def densify(self, entries, w: int, h: int, want_cells=False):
        e = np.ascontiguousarray(entries, np.float32).reshape(-1, 4)
        n = e.shape[0]
        field = np.zeros((h, w, 2), np.float32)
        cells = np.zeros((max(n, 1), 2), np.uint32) if want_cells else None
        self._check(self._lib.ofps_hip_densify(self._h, _fp(e), n, w, h, _fp(field),
                                               cells.ctypes.data_as(C.POINTER(C.c_uint32)) if want_cells else None))
        return (field, cells[:n]) if want_cells else field

    def densify_weighted(self, entries, weights, w: int, h: int, want_cells=False):
        """add_vector_weighted for every entry (motion_field.rs:164-178)."""
        e = np.ascontiguousarray(entries, np.float32).reshape(-1, 4)
        wg = np.ascontiguousarray(weights, np.float32).reshape(-1)
        assert wg.shape[0] == e.shape[0]
        field = np.zeros((h, w, 2), np.float32)
        cells = np.zeros((max(e.shape[0], 1), 2), np.uint32) if want_cells else None
        self._check(self._lib.ofps_hip_densify_weighted(self._h, _fp(e), _fp(wg), e.shape[0], w, h, _fp(field),
                                                        cells.ctypes.data_as(C.POINTER(C.c_uint32)) if want_cells else None))
        return (field, cells[:e.shape[0]]) if want_cells else field

    def densify_dev(self, d_entries: int, n_per_item: int, batch: int, w: int, h: int, d_out_field: int,
                    d_out_cells: int | None = None):
        self._check(self._lib.ofps_hip_densify_dev(self._h, C.c_void_p(d_entries), n_per_item, batch, w, h,
                                                   C.c_void_p(d_out_field), C.c_void_p(d_out_cells or 0)))


    def densify_raster_dev(self, d_entries: int, d_mask: int | None, W: int, H: int, w: int, h: int, d_out_field: int,
                           verify: bool = False):
        """Densify of a per-pixel raster producer's records (cv-decoder/src/lib.rs:239-291): one launch, no sort."""
        self._check(self._lib.ofps_hip_densify_raster_dev(self._h, C.c_void_p(d_entries), C.c_void_p(d_mask or 0), W, H, w, h,
                                                          C.c_void_p(d_out_field), 1 if verify else 0))

    def densify_to_entries(self, entries, w: int, h: int) -> np.ndarray:
        e = np.ascontiguousarray(entries, np.float32).reshape(-1, 4)
        out = np.zeros((w * h, 4), np.float32)
        n_out = C.c_size_t(0)
        self._check(self._lib.ofps_hip_densify_to_entries(self._h, _fp(e), e.shape[0], w, h, _fp(out), C.byref(n_out)))
        return out[:n_out.value].copy()

    def densify_interpolated(self, entries, w: int, h: int) -> np.ndarray:
        e = np.ascontiguousarray(entries, np.float32).reshape(-1, 4)
        field = np.zeros((h, w, 2), np.float32)
        self._check(self._lib.ofps_hip_densify_interpolated(self._h, _fp(e), e.shape[0], w, h, _fp(field)))
        return field

    # ---- A5
    def block_dim(self, min_size: float, subdivide: int) -> int:
        return int(self._lib.ofps_hip_block_dim(min_size, subdivide))

    def detect(self, entries, min_size=0.05, subdivide=3, target_motion=0.003):
        e = np.ascontiguousarray(entries, np.float32).reshape(-1, 4)
        dim = self.block_dim(min_size, subdivide)
        field = np.zeros((max(dim, 1), max(dim, 1), 2), np.float32)
        has = C.c_int(0); area = C.c_size_t(0); odim = C.c_int(0)
        self._check(self._lib.ofps_hip_detect(self._h, _fp(e), e.shape[0], min_size, subdivide, target_motion,
                                              C.byref(has), C.byref(area), C.byref(odim), _fp(field)))
        return (int(area.value), field) if has.value else None

    def detect_dev(self, d_entries: int, n_per_item: int, batch: int, min_size: float, subdivide: int,
                   target_motion: float, d_out_result: int, d_out_field: int):
        self._check(self._lib.ofps_hip_detect_dev(self._h, C.c_void_p(d_entries), n_per_item, batch, min_size, subdivide,
                                                  target_motion, C.c_void_p(d_out_result), C.c_void_p(d_out_field)))

    # ---- A6-A12
    def almeida(self, entries, aspect: float, fov_y_deg: float, use_ransac=False, num_iters=200, inlier_deg=0.05,
                num_samples=1000, seed=0):
        e = np.ascontiguousarray(entries, np.float32).reshape(-1, 4)
        q = np.zeros(4, np.float32); tr = np.zeros(3, np.float32)
        self._check(self._lib.ofps_hip_almeida(self._h, _fp(e), e.shape[0], aspect, fov_y_deg, int(use_ransac),
                                               num_iters, inlier_deg, num_samples, seed, _fp(q), _fp(tr)))
        return q, tr

    def almeida_dev(self, d_entries: int, n_per_item: int, batch: int, aspect: float, fov_y_deg: float,
                    use_ransac: bool, num_iters: int, inlier_deg: float, num_samples: int, seed: int, d_out_quat: int):
        self._check(self._lib.ofps_hip_almeida_dev(self._h, C.c_void_p(d_entries), n_per_item, batch, aspect, fov_y_deg,
                                                   int(use_ransac), num_iters, inlier_deg, num_samples, seed,
                                                   C.c_void_p(d_out_quat)))


    def checksum_dev(self, d_data: int, bytes_per_item: int, batch: int, d_out_u64: int):
        """Per-item wrapping u64 sum of device-resident data (ofps_hip_checksum_dev); enqueue only."""
        self._check(self._lib.ofps_hip_checksum_dev(self._h, C.c_void_p(d_data), bytes_per_item, batch, C.c_void_p(d_out_u64)))

    # ---- fused per-frame path
    def reset_frames(self):
        self._check(self._lib.ofps_hip_reset_frames(self._h))

    def stage_frame(self, luma: np.ndarray):
        """Upload a frame as the stream's newest frame without computing anything (a Decoder's skipped frames)."""
        luma = np.ascontiguousarray(luma, np.uint8)
        H, W = luma.shape
        self._check(self._lib.ofps_hip_stage_frame(self._h, luma.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, W))

    def pinned_frame(self, height: int, width: int) -> np.ndarray:
        """uint8[height, width] in page-locked host memory (freed with the context): frames written into it cross
        PCIe by DMA without a staging copy."""
        p = C.c_void_p(0)
        self._check(self._lib.ofps_hip_host_alloc(self._h, height * width, C.byref(p)))
        self._pinned.append(p.value)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(height, width))

    def free_pinned(self, buf: np.ndarray):
        """Releases a buffer returned by pinned_frame / pinned_array before the context is closed."""
        p = buf.ctypes.data
        if p in self._pinned:
            self._pinned.remove(p)
            self._check(self._lib.ofps_hip_host_free(self._h, C.c_void_p(p)))

    def pinned_array(self, shape, dtype=np.float32) -> np.ndarray:
        """Page-locked host array (results of push_frame_async land in it by DMA)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p(0)
        self._check(self._lib.ofps_hip_host_alloc(self._h, max(n, 16), C.byref(p)))
        self._pinned.append(p.value)
        return np.frombuffer((C.c_uint8 * max(n, 16)).from_address(p.value), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def _frame_params(self, block, search_range, detector, min_size, subdivide, target_motion, estimator, aspect, fov_y_deg,
                      use_ransac, num_iters, inlier_deg, num_samples, seed):
        return _lib.FrameParams(block, search_range, int(detector), min_size, subdivide, target_motion, int(estimator),
                                aspect, fov_y_deg, int(use_ransac), num_iters, inlier_deg, num_samples, seed)

    def push_frame_async(self, luma: np.ndarray, block=16, search_range=16, detector=True, min_size=0.05, subdivide=3,
                         target_motion=0.003, estimator=True, aspect=16 / 9, fov_y_deg=39.6 * 9 / 16, use_ransac=False,
                         num_iters=200, inlier_deg=0.05, num_samples=1000, seed=0, out_entries: np.ndarray | None = None,
                         out_field: np.ndarray | None = None) -> int:
        """Enqueues one frame (ofps_hip_push_frame_async) -> ticket.  `luma`, `out_entries` and `out_field` must stay alive
        and untouched until frame_wait(ticket) returns; at most two tickets in flight."""
        assert luma.dtype == np.uint8 and luma.ndim == 2 and luma.flags["C_CONTIGUOUS"]
        H, W = luma.shape
        prm = self._frame_params(block, search_range, detector, min_size, subdivide, target_motion, estimator, aspect, fov_y_deg,
                                 use_ransac, num_iters, inlier_deg, num_samples, seed)
        t = C.c_int(0)
        self._check(self._lib.ofps_hip_push_frame_async(self._h, luma.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, W, C.byref(prm),
                                                        _fp(out_entries) if out_entries is not None else None,
                                                        _fp(out_field) if out_field is not None else None, C.byref(t)))
        return int(t.value)

    def push_frames_async(self, frames: np.ndarray, block=16, search_range=16, detector=True, min_size=0.05, subdivide=3,
                          target_motion=0.003, estimator=True, aspect=16 / 9, fov_y_deg=39.6 * 9 / 16, use_ransac=False,
                          num_iters=200, inlier_deg=0.05, num_samples=1000, seed=0, out_entries: np.ndarray | None = None) -> int:
        """Batched form (ofps_hip_push_frames_async): frames uint8 [n, H, W] contiguous -> ticket; keep `frames` and
        `out_entries` ([n, nblk, 4] float32) alive until frames_wait(ticket)."""
        assert frames.dtype == np.uint8 and frames.ndim == 3 and frames.flags["C_CONTIGUOUS"]
        n, H, W = frames.shape
        prm = self._frame_params(block, search_range, detector, min_size, subdivide, target_motion, estimator, aspect, fov_y_deg,
                                 use_ransac, num_iters, inlier_deg, num_samples, seed)
        t = C.c_int(0)
        self._check(self._lib.ofps_hip_push_frames_async(self._h, frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, W, H, W, W * H, C.byref(prm),
                                                         _fp(out_entries) if out_entries is not None else None, C.byref(t)))
        self._batch_n = getattr(self, "_batch_n", {})
        self._batch_n[t.value] = n
        return t.value

    def frames_wait(self, ticket: int) -> list:
        n = self._batch_n.pop(ticket)
        res = (_lib.FrameResult * n)()
        self._check(self._lib.ofps_hip_frames_wait(self._h, ticket, res))
        return [{"have_vectors": bool(r.have_vectors), "n_vectors": int(r.n_vectors),
                 "motion": (int(r.area), int(r.dim)) if r.has_motion else None, "quat": np.array(list(r.quat), np.float32)} for r in res]

    def frame_wait(self, ticket: int) -> dict:
        res = _lib.FrameResult()
        self._check(self._lib.ofps_hip_frame_wait(self._h, ticket, C.byref(res)))
        return {"have_vectors": bool(res.have_vectors), "n_vectors": int(res.n_vectors),
                "motion": (int(res.area), int(res.dim)) if res.has_motion else None, "quat": np.array(list(res.quat), np.float32)}

    def push_frame(self, luma: np.ndarray, block=16, search_range=16, detector=True, min_size=0.05, subdivide=3,
                   target_motion=0.003, estimator=True, aspect=16 / 9, fov_y_deg=39.6 * 9 / 16, use_ransac=False,
                   num_iters=200, inlier_deg=0.05, num_samples=1000, seed=0, want_entries=False, want_field=False):
        """-> dict(have_vectors, n_vectors, motion=None|(area, field|None), quat, entries|None)"""
        luma = np.ascontiguousarray(luma, np.uint8)
        H, W = luma.shape
        prm = _lib.FrameParams(block, search_range, int(detector), min_size, subdivide, target_motion, int(estimator),
                               aspect, fov_y_deg, int(use_ransac), num_iters, inlier_deg, num_samples, seed)
        res = _lib.FrameResult()
        nb = int(self._lib.ofps_hip_sad_block_count(W, H, block))
        ent = np.zeros((max(nb, 1), 4), np.float32) if want_entries else None
        dim = self.block_dim(min_size, subdivide)
        fld = np.zeros((dim, dim, 2), np.float32) if want_field else None
        self._check(self._lib.ofps_hip_push_frame(self._h, luma.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, W, C.byref(prm),
                                                  C.byref(res), _fp(ent) if want_entries else None,
                                                  _fp(fld) if want_field else None))
        motion = (int(res.area), fld) if res.has_motion else None
        return {"have_vectors": bool(res.have_vectors), "n_vectors": int(res.n_vectors), "motion": motion,
                "quat": np.array(list(res.quat), np.float32), "entries": ent[:nb] if (want_entries and res.have_vectors) else None}


def device_count() -> int:
    return int(_lib.load().ofps_hip_device_count())


class MultiDevice:
    """In-process multi-device dispatcher (include/ofps_hip.h: ofps_hip_multi_*): one worker thread + one context per entry
    of `devices` (entries may repeat), frame pairs split into contiguous ranges, results in pair order."""

    def __init__(self, devices):
        self._lib = _lib.load()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self._lib.ofps_hip_multi_init(devs, len(devices), C.byref(h))
        if rc != 0:
            raise OfpsHipError(rc, (self._lib.ofps_hip_multi_last_error(None) or b"").decode())
        self._h = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ofps_hip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise OfpsHipError(rc, (self._lib.ofps_hip_multi_last_error(self._h) or b"").decode())

    @staticmethod
    def pair_range(n_pairs: int, n_workers: int, k: int):
        f, c = C.c_size_t(0), C.c_size_t(0)
        _lib.load().ofps_hip_multi_pair_range(n_pairs, n_workers, k, C.byref(f), C.byref(c))
        return int(f.value), int(c.value)

    @staticmethod
    def frame_range(n_pairs: int, n_workers: int, k: int, ref_mode: int):
        f, c = C.c_size_t(0), C.c_size_t(0)
        _lib.load().ofps_hip_multi_frame_range(n_pairs, n_workers, k, ref_mode, C.byref(f), C.byref(c))
        return int(f.value), int(c.value)

    @staticmethod
    def stream_plan(batch: int, n_workers: int):
        """-> (worker, halo_slot, ticket_slot) of batch number `batch` of a stream (pure function, no device needed)."""
        w, h, t = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.load().ofps_hip_multi_stream_plan(batch, n_workers, C.byref(w), C.byref(h), C.byref(t))
        return int(w.value), int(h.value), int(t.value)

    def push_frames_async(self, frames: np.ndarray, block=16, search_range=16, detector=True, min_size=0.05, subdivide=3,
                          target_motion=0.003, estimator=True, aspect=16 / 9, fov_y_deg=39.6 * 9 / 16, use_ransac=False,
                          num_iters=200, inlier_deg=0.05, num_samples=1000, seed=0, out_entries: np.ndarray | None = None) -> int:
        """One batch of consecutive frames of the stream (uint8 [n, H, W] contiguous) -> ticket; keep `frames` and `out_entries`
        ([n, nblk, 4] float32) alive until frames_wait(ticket).  Batches are dealt to the workers round-robin.  A view with
        padded rows (unit stride along x, rows `stride` >= W bytes apart, frames `stride * H` or more apart) is passed as it is."""
        assert frames.dtype == np.uint8 and frames.ndim == 3
        n, H, W = frames.shape
        pitch, stride, unit = frames.strides
        assert unit == 1 and stride >= W and (n == 1 or pitch >= stride * H), "frames: rows must be byte-contiguous"
        prm = _lib.FrameParams(block, search_range, int(detector), min_size, subdivide, target_motion, int(estimator),
                               aspect, fov_y_deg, int(use_ransac), num_iters, inlier_deg, num_samples, seed)
        t = C.c_int(0)
        self._check(self._lib.ofps_hip_multi_push_frames_async(self._h, C.cast(C.c_void_p(frames.ctypes.data), C.POINTER(C.c_uint8)), n, W, H, stride,
                                                               max(pitch, stride * H), C.byref(prm),
                                                               _fp(out_entries) if out_entries is not None else None, C.byref(t)))
        self._batch_n = getattr(self, "_batch_n", {})
        self._batch_n[t.value] = n
        return t.value

    def frames_wait(self, ticket: int) -> list:
        n = getattr(self, "_batch_n", {}).pop(ticket, 1)
        res = (_lib.FrameResult * n)()
        self._check(self._lib.ofps_hip_multi_frames_wait(self._h, ticket, res))
        return [{"have_vectors": bool(r.have_vectors), "n_vectors": int(r.n_vectors),
                 "motion": (int(r.area), int(r.dim)) if r.has_motion else None, "quat": np.array(list(r.quat), np.float32)} for r in res]

    def reset_frames(self):
        self._check(self._lib.ofps_hip_multi_reset_frames(self._h))

    def sad_flow(self, frames: np.ndarray, block: int, search_range: int, ref_mode: int = 0) -> np.ndarray:
        """frames: uint8 [n_frames, H, stride>=W is the array's row pitch] -> entries [n_frames-1, nblk, 4]."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, H, W = frames.shape
        nb = int(self._lib.ofps_hip_sad_block_count(W, H, block))
        out = np.zeros((max(n - 1, 0), nb, 4), np.float32)
        self._check(self._lib.ofps_hip_multi_sad_flow(self._h, frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, W, H, W, W * H, ref_mode,
                                                      block, search_range, _fp(out)))
        return out

    def stage_frames(self, frames: np.ndarray, ref_mode: int = 0):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, H, W = frames.shape
        self._geom = (n, H, W)
        self._check(self._lib.ofps_hip_multi_stage_frames(self._h, frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, W, H, W, W * H, ref_mode))

    def run_resident(self, block: int, search_range: int, steps: int = 1, timed: bool = False):
        """-> None, or with timed=True the per-worker HIP-event milliseconds of the `steps` launches."""
        ms = np.zeros(len(self.devices), np.float32) if timed else None
        self._check(self._lib.ofps_hip_multi_run_resident(self._h, block, search_range, steps, _fp(ms) if timed else None))
        return ms

    def fetch(self, block: int) -> np.ndarray:
        n, H, W = self._geom
        nb = int(self._lib.ofps_hip_sad_block_count(W, H, block))
        out = np.zeros((max(n - 1, 0), nb, 4), np.float32)
        self._check(self._lib.ofps_hip_multi_fetch(self._h, block, _fp(out)))
        return out
