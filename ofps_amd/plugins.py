"""Python mirror of the reference's plugin interface for the hot path (same names, argument meaning and
error behaviour as ofps/src/{decoder,estimator,detection}.rs and ofps/src/plugins/properties.rs), on top of the
C ABI.  It exists so tests and scripts read like the reference's own code; every result comes from
libofps_hip.so.  The C++ twin is ofps_amd/host/ofps_host.hpp; the Rust shim is in INTEGRATION.md.
"""
from __future__ import annotations

import math
from typing import BinaryIO, Iterable, Iterator, Optional

import numpy as np

from .runtime import HipContext


class StandardCamera:
    """ofps/src/camera.rs:12-35,166-177 -- parameters only; the projection math runs on the device."""

    def __init__(self, aspect: float, fov_y: float):
        self.aspect, self.fov_y = float(aspect), float(fov_y)

    def aspect_ratio(self) -> float:
        return self.aspect

    def fov(self):
        ty = math.tan(math.radians(self.fov_y) / 2.0)
        return (math.degrees(math.atan(self.aspect * ty)) * 2.0, self.fov_y)


class MotionField:
    """ofps/src/motion_field.rs:7-115: fixed-size field, cell (x, y) stored at width*y + x."""

    def __init__(self, data: np.ndarray):
        self.vf = np.asarray(data, np.float32)            # [h, w, 2]

    def dim(self):
        return (self.vf.shape[1], self.vf.shape[0])

    def size(self) -> int:
        return self.vf.shape[0] * self.vf.shape[1]

    def as_slice(self) -> np.ndarray:
        return self.vf.reshape(-1)

    def get_motion(self, x: int, y: int) -> np.ndarray:
        return self.vf[y, x]

    def iter(self):
        h, w = self.vf.shape[:2]
        return ((x, y, self.vf[y, x]) for y in range(h) for x in range(w))


class Properties:
    """plugins/properties.rs:6-18: props() -> [(name, kind, value, min, max)], set_prop(name, value)."""
    _PROPS: tuple = ()

    def props(self):
        return [(name, kind, getattr(self, attr), lo, hi) for name, kind, attr, lo, hi in self._PROPS]

    def set_prop(self, name: str, value) -> bool:
        for pname, kind, attr, lo, hi in self._PROPS:
            if pname == name:
                setattr(self, attr, {"bool": bool, "float": float, "usize": int}[kind](value))
                return True
        return False


class HipSadDecoder(Properties):
    """Decoder (ofps/src/decoder.rs:45-73) over an iterator / raw stream of luma frames: full-search SAD
    block vectors in av-decoder's record convention (av-decoder/src/lib.rs:404-419)."""
    _PROPS = (("Block size", "usize", "block", 8, 16), ("Search range", "usize", "range", 8, 32),
              ("Exact pruning", "bool", "pruned", None, None))      # same vectors; faster on smooth camera motion (16x16, +-16)

    def __init__(self, frames: Iterable[np.ndarray], framerate: Optional[float] = None, device: int = 0):
        self.ctx = HipContext(device)
        self._it: Iterator[np.ndarray] = iter(frames)
        self.block, self.range = 16, 16
        self.pruned = False
        self._prev: Optional[np.ndarray] = None
        self._cur: Optional[np.ndarray] = None
        self._fps = framerate

    @classmethod
    def from_raw_stream(cls, f: BinaryIO, width: int, height: int, framerate: Optional[float] = None, device: int = 0):
        def gen():
            while True:
                buf = f.read(width * height)
                if len(buf) != width * height:
                    return
                yield np.frombuffer(buf, np.uint8).reshape(height, width)
        return cls(gen(), framerate, device)

    def process_frame(self, field: list, out_frame: Optional[list] = None, skip_frames: int = 0) -> bool:
        """True: vectors appended to `field` (callers clear it); False: no vectors this frame; raises
        StopIteration-derived EOFError at end of stream (the Err that ends the reference's worker loop).
        The vectors relate the last two frames read.  Only newly read frames cross PCIe, from a page-locked
        buffer: the previous frame stays on the device (ofps_hip_stage_frame / ofps_hip_push_frame)."""
        for k in range(skip_frames + 1):
            try:
                frame = np.asarray(next(self._it), np.uint8)
            except StopIteration:
                raise EOFError("end of stream") from None
            if k < skip_frames - 1:
                continue                                   # consumed and dropped
            if self._cur is None or self._cur.shape != frame.shape:
                if self._cur is not None:
                    self.ctx.free_pinned(self._cur)        # a geometry change must not leak the old page-locked buffer
                self._cur = self.ctx.pinned_frame(*frame.shape)
            np.copyto(self._cur, frame)
            if k == skip_frames - 1:
                self.ctx.stage_frame(self._cur)            # becomes the frame the vectors are relative to
        if out_frame is not None:
            out_frame[:] = [self._cur.copy()]
        self.ctx.set_sad_mode(self.ctx.SAD_PRUNED if self.pruned else self.ctx.SAD_EXHAUSTIVE)
        r = self.ctx.push_frame(self._cur, self.block, self.range, detector=False, estimator=False, want_entries=True)
        if not r["have_vectors"]:                          # first frame of the stream / geometry change
            return False
        field.extend(r["entries"])
        return True

    def get_framerate(self):
        return self._fps

    def get_aspect(self):
        return None if self._cur is None else (self._cur.shape[1], self._cur.shape[0])


class HipLkDecoder(HipSadDecoder):
    """Decoder producing dense per-pixel flow the way cv-decoder does (cv-decoder/src/lib.rs:82-294).  "Process Fullres" = true (the
    default, as in the reference): flow on the full frames, records down-sampled through the densifier to the (Width, Height)-capped grid;
    false: every frame is resized (INTER_LINEAR) to that grid first and every unmasked pixel of the REDUCED frame is one record
    (:124-133,274-276).  Frames: [H, W] luma (this build's raw streams) or [H, W, 3 | 4] colour with `frame_format` = FMT_BGR / FMT_RGBA /
    FMT_BGRA -- converted with cvt_color(BGR2GRAY)'s formula on the device (:135).  "Fullres records" is this build's own output form (one
    record per full-resolution pixel, no down-sampling), not a cv-decoder mode."""
    _PROPS = (("Width", "usize", "max_w", 1, 2000), ("Height", "usize", "max_h", 1, 2000),
              ("Pyramid levels", "usize", "levels", 1, 8), ("Window radius", "usize", "radius", 1, 15),
              ("Iterations", "usize", "iters", 1, 64), ("Contrast mask", "bool", "contrast_mask", None, None),
              ("Process Fullres", "bool", "process_fullres", None, None), ("Fullres records", "bool", "fullres_records", None, None))

    def __init__(self, frames, framerate=None, device: int = 0, frame_format: int = HipContext.FMT_LUMA):
        super().__init__(frames, framerate, device)
        self.max_w, self.max_h, self.levels, self.radius, self.iters = 150, 150, 3, 4, 3
        self.contrast_mask, self.process_fullres = True, True          # cv-decoder's Farneback path always masks
        self.fullres_records = False
        self.frame_format = frame_format
        self._mode = None

    def get_aspect(self):
        """cv-decoder/src/lib.rs:296-298: the size of `self.gray` -- the REDUCED frame's with "Process Fullres" = false"""
        if self._cur is None:
            return None
        H, W = self._cur.shape[:2]
        return (W, H) if self.process_fullres else self.ctx.cv_grid(W, H, self.max_w, self.max_h)

    def _shown_frame(self) -> np.ndarray:
        """what cv-decoder hands out as the frame (cv-decoder/src/lib.rs:144-154: `self.frame`): the arriving frame, or -- "Process Fullres" =
        false -- the frame resized to the capped grid (channels kept)"""
        if self.process_fullres:
            return self._cur
        H, W = self._cur.shape[:2]
        gw, gh = self.ctx.cv_grid(W, H, self.max_w, self.max_h)
        return self.ctx.resize_linear(self._cur, gw, gh, self.frame_format)

    def _flow_kw(self) -> dict:
        return dict(contrast_mask=self.contrast_mask, reduced=not self.process_fullres, fmt=self.frame_format,
                    fullres_records=self.fullres_records and self.process_fullres)

    def _decode(self, field: list, kw: dict) -> bool:
        """the pair (self._prev, self._cur) through the library's stream form; the frame on the device from the last call is this call's
        previous frame unless frames were skipped or the mode changed: then (and on the first pair) the previous frame goes up first"""
        mode = tuple(sorted(kw.items())) + (self.levels, self.radius, self.iters, self.max_w, self.max_h)
        if getattr(self, "_on_device", None) is not self._prev or mode != self._mode:
            # (frames were skipped: the stream's last flow stays the initial flow, cv-decoder's self.flow persists, cv-decoder/src/lib.rs:161-165)
            self.ctx.lk_reset() if (mode != self._mode or getattr(self, "_on_device", None) is None) else self.ctx.lk_rewind()
            self.ctx.lk_push_frame(self._prev, self.levels, self.radius, self.iters, self.max_w, self.max_h, **kw)
        ent, _ = self.ctx.lk_push_frame(self._cur, self.levels, self.radius, self.iters, self.max_w, self.max_h, **kw)
        self._on_device, self._mode = self._cur, mode
        field.extend(ent)
        return True

    def process_frame(self, field: list, out_frame=None, skip_frames: int = 0) -> bool:
        for _ in range(skip_frames + 1):
            self._prev = self._cur
            try:
                self._cur = np.ascontiguousarray(next(self._it), np.uint8)
            except StopIteration:
                raise EOFError("failed to grab frame") from None
        if out_frame is not None:
            out_frame[:] = [self._shown_frame()]
        if self._prev is None or self._prev.shape != self._cur.shape:
            self._on_device = None
            return False
        return self._decode(field, self._flow_kw())


class HipFlowDecoder(HipLkDecoder):
    """The dense decoder in the reference's own algorithm family: Farneback's polynomial-expansion flow with cv-decoder's arguments
    (cv-decoder/src/lib.rs:188-199: levels 5, winsize 13, iterations 3, poly_n 7, poly_sigma 1.5), then cv-decoder's contrast mask and
    down-sampling exactly as HipLkDecoder (ofps_amd/csrc/farneback.hip; "Window radius" r means winsize 2 r + 1)."""

    _PROPS = tuple(p if p[0] != "Window radius" else ("Window radius", "usize", "radius", 1, 7) for p in HipLkDecoder._PROPS)   # winsize <= 15

    def __init__(self, frames, framerate=None, device: int = 0, frame_format: int = HipContext.FMT_LUMA):
        super().__init__(frames, framerate, device, frame_format)
        self.levels, self.radius, self.iters = 5, 6, 3
        self.use_previous_flow = True      # OPTFLOW_USE_INITIAL_FLOW as cv-decoder sets it (the library keeps the flow; a stream restart forgets it)

    def process_frame(self, field: list, out_frame=None, skip_frames: int = 0) -> bool:
        for _ in range(skip_frames + 1):
            self._prev = self._cur
            try:
                self._cur = np.ascontiguousarray(next(self._it), np.uint8)
            except StopIteration:
                raise EOFError("failed to grab frame") from None
        if out_frame is not None:
            out_frame[:] = [self._shown_frame()]
        if self._prev is None or self._prev.shape != self._cur.shape:
            self._on_device = None
            return False
        # cv-decoder/src/lib.rs:161-165: from its second pair on the decoder passes its previous flow as the initial flow
        return self._decode(field, dict(self._flow_kw(), farneback=True, use_previous=self.use_previous_flow))


class HipBlockMotionDetection(Properties):
    """Detector (ofps/src/detection.rs:11; block-motion-detector/src/lib.rs:13-46)."""
    _PROPS = (("Min size", "float", "min_size", 0.01, 1.0), ("Subdivisions", "usize", "subdivide", 1, 16),
              ("Target motion", "float", "target_motion", 0.0001, 0.1))

    def __init__(self, device: int = 0):
        self.ctx = HipContext(device)
        self.min_size, self.subdivide, self.target_motion = 0.05, 3, 0.003

    def detect_motion(self, motion):
        """-> None | (area, MotionField)"""
        r = self.ctx.detect(np.asarray(motion, np.float32).reshape(-1, 4), self.min_size, self.subdivide, self.target_motion)
        return None if r is None else (r[0], MotionField(r[1]))


class HipAlmeidaEstimator(Properties):
    """Estimator (ofps/src/estimator.rs:8-53; almeida-estimator/src/lib.rs:57-121)."""
    _PROPS = (("Use ransac", "bool", "use_ransac", None, None), ("Ransac iters", "usize", "num_iters", 1, 500),
              ("Inlier threshold", "float", "inlier_angle", 0.01, 1.0), ("Ransac samples", "usize", "ransac_samples", 100, 16000))

    def __init__(self, device: int = 0):
        self.ctx = HipContext(device)
        self.use_ransac, self.num_iters, self.inlier_angle, self.ransac_samples = True, 200, 0.05, 1000
        self.seed = 0

    def estimate(self, motion_vectors, camera: StandardCamera, move_magnitude=None):
        """-> (quaternion (w,i,j,k), translation (0,0,0))"""
        self.seed += 1
        return self.ctx.almeida(np.asarray(motion_vectors, np.float32).reshape(-1, 4), camera.aspect_ratio(), camera.fov()[1],
                                self.use_ransac, self.num_iters, self.inlier_angle, self.ransac_samples, self.seed)

    def motion_step(self, motion_vectors, camera, move_magnitude, rot, pos):
        """estimator.rs:38-53: pos += rot * tr; rot = r * rot (tr is always zero for this estimator)."""
        r, tr = self.estimate(motion_vectors, camera, move_magnitude)
        return quat_mul(r, rot), np.asarray(pos, np.float32) + np.asarray(tr, np.float32)


def quat_mul(a, b) -> np.ndarray:
    aw, ai, aj, ak = [float(v) for v in a]
    bw, bi, bj, bk = [float(v) for v in b]
    return np.array([aw * bw - ai * bi - aj * bj - ak * bk, aw * bi + ai * bw + aj * bk - ak * bj,
                     aw * bj - ai * bk + aj * bw + ak * bi, aw * bk + ai * bj - aj * bi + ak * bw], np.float32)
