"""ctypes front-end for the CPU oracle (oracle/ofps_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package `ofps_amd` never imports this module.  See
ofps_oracle.h for the parity status of each restated function.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libofps_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (see oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("ofps_oracle.c", "farneback_oracle.c", "frontend_oracle.c", "ofps_oracle.h")]
    stale = not os.path.exists(_LIB_PATH) or any(os.path.getmtime(_LIB_PATH) < os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libofps_oracle.so"])
    return _LIB_PATH


class Camera(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("aspect", "fov_y", "m00", "m11", "m22", "m23", "r00", "r11", "r32", "r33")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_camera_new.argtypes = [C.POINTER(Camera), C.c_float, C.c_float]
        L.orc_camera_delta.argtypes = [C.POINTER(Camera), fp, fp, fp]
        L.orc_camera_unproject.argtypes = [C.POINTER(Camera), fp, fp, fp]
        L.orc_camera_project.argtypes = [C.POINTER(Camera), fp, fp, fp]
        L.orc_camera_point_angle.argtypes = [C.POINTER(Camera), fp, fp]
        L.orc_mat4_from_euler.argtypes = [C.c_float, C.c_float, C.c_float, fp]
        L.orc_mat4_look_at_rh.argtypes = [fp, fp, fp, fp]
        L.orc_quat_from_euler.argtypes = [C.c_float, C.c_float, C.c_float, fp]
        L.orc_quat_mul.argtypes = [fp, fp, fp]
        L.orc_quat_to_homogeneous.argtypes = [fp, fp]
        L.orc_quat_transform_vector.argtypes = [fp, fp, fp]
        L.orc_quat_angle_to.argtypes = [fp, fp]
        L.orc_quat_angle_to.restype = C.c_float
        L.orc_lu3_solve.argtypes = [fp, fp, fp]
        L.orc_lu3_solve.restype = C.c_int
        L.orc_densify.argtypes = [fp, C.c_size_t, C.c_int, C.c_int, fp, C.POINTER(C.c_uint32), fp]
        L.orc_densify_weighted.argtypes = [fp, fp, C.c_size_t, C.c_int, C.c_int, fp, C.POINTER(C.c_uint32)]
        L.orc_densify_to_entries.argtypes = [fp, C.c_size_t, C.c_int, C.c_int, fp]
        L.orc_densify_to_entries.restype = C.c_size_t
        L.orc_densify_interpolated.argtypes = [fp, C.c_size_t, C.c_int, C.c_int, fp]
        L.orc_block_dim.argtypes = [C.c_float, C.c_size_t]
        L.orc_block_dim.restype = C.c_int
        L.orc_detect_motion.argtypes = [fp, C.c_size_t, C.c_float, C.c_size_t, C.c_float,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_int), fp]
        L.orc_detect_motion.restype = C.c_int
        L.orc_solve_ypr_given.argtypes = [fp, C.c_size_t, C.POINTER(Camera), fp]
        L.orc_solve_ypr_given_ex.argtypes = [fp, C.c_size_t, C.POINTER(Camera), C.c_float, C.c_size_t, C.c_int, fp, fp]
        L.orc_solve_ypr_given_mt.argtypes = [fp, C.c_size_t, C.POINTER(Camera), C.c_int, fp]
        L.orc_solve_ypr_ransac_mt.argtypes = [fp, C.c_size_t, C.POINTER(Camera), C.c_size_t, C.c_float, C.c_size_t, C.c_uint64,
                                              C.c_int, fp]
        L.orc_quat_inverse.argtypes = [fp, fp]
        L.orc_almeida_model.argtypes = [fp, C.c_size_t, C.POINTER(Camera), fp, fp]
        L.orc_solve_ypr_ransac.argtypes = [fp, C.c_size_t, C.POINTER(Camera), C.c_size_t, C.c_float,
                                           C.c_size_t, C.c_uint64, fp, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_size_t)]
        L.orc_sample_index.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_sample_index.restype = C.c_uint32
        L.orc_sad_flow.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, fp, C.POINTER(C.c_int32), C.c_int]
        L.orc_sad_flow.restype = C.c_size_t
        L.orc_sad_flow_ex.argtypes = L.orc_sad_flow.argtypes + [C.c_int]
        L.orc_sad_flow_ex.restype = C.c_size_t
        L.orc_lk_flow.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, fp]
        L.orc_lk_flow.restype = C.c_int
        L.orc_lk_flow_init.argtypes = L.orc_lk_flow.argtypes[:-1] + [fp, fp]
        L.orc_lk_flow_init.restype = C.c_int
        L.orc_lk_flow_trace.argtypes = L.orc_lk_flow_init.argtypes + [fp]
        L.orc_lk_flow_trace.restype = C.c_int
        L.orc_flow_to_entries.argtypes = [fp, C.c_int, C.c_int, fp]
        L.orc_contrast_mask.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.orc_masked_flow_to_entries.argtypes = [fp, C.POINTER(C.c_uint8), C.c_int, C.c_int, fp]
        L.orc_masked_flow_to_entries.restype = C.c_size_t
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads.restype = C.c_int
        L.orc_sad_simd_level.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def camera(aspect: float, fov_y_deg: float) -> Camera:
    c = Camera()
    lib().orc_camera_new(C.byref(c), aspect, fov_y_deg)
    return c


def camera_delta(cam: Camera, pos, rot4) -> np.ndarray:
    p = _f32(pos); r = _f32(rot4).reshape(16); o = np.zeros(2, np.float32)
    lib().orc_camera_delta(C.byref(cam), _fp(p), _fp(r), _fp(o))
    return o


def camera_unproject(cam: Camera, pos, inv_view4) -> np.ndarray:
    p = _f32(pos); v = _f32(inv_view4).reshape(16); o = np.zeros(3, np.float32)
    lib().orc_camera_unproject(C.byref(cam), _fp(p), _fp(v), _fp(o))
    return o


def camera_project(cam: Camera, world, view4) -> np.ndarray:
    w = _f32(world); v = _f32(view4).reshape(16); o = np.zeros(2, np.float32)
    lib().orc_camera_project(C.byref(cam), _fp(w), _fp(v), _fp(o))
    return o


def camera_point_angle(cam: Camera, pos) -> np.ndarray:
    p = _f32(pos); o = np.zeros(2, np.float32)
    lib().orc_camera_point_angle(C.byref(cam), _fp(p), _fp(o))
    return o


def mat4_from_euler(roll, pitch, yaw) -> np.ndarray:
    o = np.zeros(16, np.float32)
    lib().orc_mat4_from_euler(roll, pitch, yaw, _fp(o))
    return o.reshape(4, 4)


def look_at_rh(eye, target, up) -> np.ndarray:
    o = np.zeros(16, np.float32)
    e, t, u = _f32(eye), _f32(target), _f32(up)
    lib().orc_mat4_look_at_rh(_fp(e), _fp(t), _fp(u), _fp(o))
    return o.reshape(4, 4)


def quat_from_euler(roll, pitch, yaw) -> np.ndarray:
    o = np.zeros(4, np.float32)
    lib().orc_quat_from_euler(roll, pitch, yaw, _fp(o))
    return o


def quat_to_homogeneous(q) -> np.ndarray:
    q = _f32(q); o = np.zeros(16, np.float32)
    lib().orc_quat_to_homogeneous(_fp(q), _fp(o))
    return o.reshape(4, 4)


def quat_transform_vector(q, v) -> np.ndarray:
    q = _f32(q); v = _f32(v); o = np.zeros(3, np.float32)
    lib().orc_quat_transform_vector(_fp(q), _fp(v), _fp(o))
    return o


def quat_angle_to(a, b) -> float:
    a = _f32(a); b = _f32(b)
    return float(lib().orc_quat_angle_to(_fp(a), _fp(b)))


def lu3_solve(a3, b3):
    a = _f32(a3).reshape(9); b = _f32(b3); x = np.zeros(3, np.float32)
    ok = lib().orc_lu3_solve(_fp(a), _fp(b), _fp(x))
    return (x if ok else None)


def densify(entries, w: int, h: int, want_cells: bool = False, want_counts: bool = False):
    e = _f32(entries).reshape(-1, 4); n = e.shape[0]
    field = np.zeros((h, w, 2), np.float32)
    cells = np.zeros((max(n, 1), 2), np.uint32) if want_cells else None
    counts = np.zeros((h, w, 2), np.float32) if want_counts else None
    lib().orc_densify(_fp(e), n, w, h, _fp(field),
                      cells.ctypes.data_as(C.POINTER(C.c_uint32)) if want_cells else None,
                      _fp(counts) if want_counts else None)
    out = [field]
    if want_cells:
        out.append(cells[:n])
    if want_counts:
        out.append(counts)
    return out[0] if len(out) == 1 else tuple(out)


def densify_weighted(entries, weights, w: int, h: int) -> np.ndarray:
    e = _f32(entries).reshape(-1, 4); wg = _f32(weights).reshape(-1)
    field = np.zeros((h, w, 2), np.float32)
    lib().orc_densify_weighted(_fp(e), _fp(wg), e.shape[0], w, h, _fp(field), None)
    return field


def densify_to_entries(entries, w: int, h: int) -> np.ndarray:
    e = _f32(entries).reshape(-1, 4); n = e.shape[0]
    out = np.zeros((w * h, 4), np.float32)
    k = lib().orc_densify_to_entries(_fp(e), n, w, h, _fp(out))
    return out[:k].copy()


def densify_interpolated(entries, w: int, h: int) -> np.ndarray:
    e = _f32(entries).reshape(-1, 4); n = e.shape[0]
    field = np.zeros((h, w, 2), np.float32)
    lib().orc_densify_interpolated(_fp(e), n, w, h, _fp(field))
    return field


def block_dim(min_size: float, subdivide: int) -> int:
    return int(lib().orc_block_dim(min_size, subdivide))


def detect_motion(entries, min_size=0.05, subdivide=3, target_motion=0.003):
    """-> None or (area, field[dim,dim,2]) exactly like Detector::detect_motion."""
    e = _f32(entries).reshape(-1, 4); n = e.shape[0]
    dim = block_dim(min_size, subdivide)
    field = np.zeros((dim, dim, 2), np.float32)
    area = C.c_size_t(0); odim = C.c_int(0)
    some = lib().orc_detect_motion(_fp(e), n, min_size, subdivide, target_motion,
                                   C.byref(area), C.byref(odim), _fp(field))
    assert odim.value == dim
    return (int(area.value), field) if some else None


def solve_ypr_given(entries, cam: Camera, threads: int = 1) -> np.ndarray:
    """threads > 1: the per-vector loop and the twelve dot products on that many host threads -- same bits."""
    e = _f32(entries).reshape(-1, 4); q = np.zeros(4, np.float32)
    if threads > 1:
        lib().orc_solve_ypr_given_mt(_fp(e), e.shape[0], C.byref(cam), threads, _fp(q))
    else:
        lib().orc_solve_ypr_given(_fp(e), e.shape[0], C.byref(cam), _fp(q))
    return q


def solve_ypr_given_ex(entries, cam: Camera, alpha: float, limit: int, order: int = 0):
    """-> (q, steps[limit, 4]): the solver with ALPHA / step count / composition order as arguments (ofps_oracle.h);
    steps[i] = cumulative point rotation after step i (not inverted)."""
    e = _f32(entries).reshape(-1, 4); q = np.zeros(4, np.float32)
    steps = np.zeros((max(limit, 1), 4), np.float32)
    lib().orc_solve_ypr_given_ex(_fp(e), e.shape[0], C.byref(cam), alpha, limit, order, _fp(steps), _fp(q))
    return q, steps[:limit]


def almeida_model(entries, cam: Camera, rotation=(1.0, 0.0, 0.0, 0.0)) -> np.ndarray:
    """One pass of the solver's loop body (almeida-estimator/src/lib.rs:140-183): the raw LU solution, units of EPS."""
    e = _f32(entries).reshape(-1, 4); r = _f32(rotation); m = np.zeros(3, np.float32)
    lib().orc_almeida_model(_fp(e), e.shape[0], C.byref(cam), _fp(r), _fp(m))
    return m


def quat_mul(a, b) -> np.ndarray:
    a = _f32(a); b = _f32(b); o = np.zeros(4, np.float32)
    lib().orc_quat_mul(_fp(a), _fp(b), _fp(o))
    return o


def solve_ypr_ransac(entries, cam: Camera, num_iters=200, inlier_deg=0.05, num_samples=1000,
                     seed=0, want_inliers=False, threads: int = 1):
    """threads > 1: the hypotheses on that many host threads (first maximum wins, as in the loop) -- same bits."""
    e = _f32(entries).reshape(-1, 4); q = np.zeros(4, np.float32)
    if threads > 1:
        assert not want_inliers
        lib().orc_solve_ypr_ransac_mt(_fp(e), e.shape[0], C.byref(cam), num_iters, inlier_deg, num_samples, seed, threads, _fp(q))
        return q
    inl = np.zeros(max(num_samples, 1), np.uint32); n_inl = C.c_size_t(0)
    lib().orc_solve_ypr_ransac(_fp(e), e.shape[0], C.byref(cam), num_iters, inlier_deg, num_samples,
                               seed, _fp(q), inl.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_inl))
    return (q, inl[:n_inl.value].copy()) if want_inliers else q


def lk_spec_revision() -> int:
    """2 = fused multiply-adds in the bilinear sample and the residual sums (ofps_oracle.c: ORC_LK_SPEC_FMA)."""
    return int(lib().orc_lk_spec_revision())


def sample_index(seed, it, stream, i, n) -> int:
    return int(lib().orc_sample_index(seed, it, stream, i, n))


def sad_flow(prev, cur, B: int, R: int, threads: int = 1, stride=None, simd: bool = True):
    """-> (entries[nblk,4] f32, best[nblk,3] int32 (dx,dy,sad))"""
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    H, Wp = prev.shape
    W = Wp if stride is None else stride[0]
    st = Wp
    nb = (W // B) * (H // B)
    ent = np.zeros((max(nb, 1), 4), np.float32); best = np.zeros((max(nb, 1), 3), np.int32)
    u8 = C.POINTER(C.c_uint8)
    k = lib().orc_sad_flow_ex(prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, st, B, R,
                              _fp(ent), best.ctypes.data_as(C.POINTER(C.c_int32)), threads, int(simd))
    assert k == nb
    return ent[:nb], best[:nb]


def lk_coarsest_shape(W: int, H: int, levels: int):
    """(h, w) of the coarsest pyramid level: W, H halved (rounding up) levels - 1 times."""
    for _ in range(1, levels):
        W, H = (W + 1) // 2, (H + 1) // 2
    return H, W


def lk_flow(prev, cur, levels: int = 3, radius: int = 4, iters: int = 3, init=None) -> np.ndarray:
    """-> flow[H, W, 2] f32 (u, v): prev(x,y) ~ cur(x+u, y+v).  init: the coarsest level's starting flow [h_L, w_L, 2]."""
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    H, W = prev.shape
    out = np.zeros((H, W, 2), np.float32)
    u8 = C.POINTER(C.c_uint8)
    if init is None:
        ok = lib().orc_lk_flow(prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W, levels, radius, iters, _fp(out))
    else:
        init = _f32(init)
        assert init.shape == lk_coarsest_shape(W, H, levels) + (2,), init.shape
        ok = lib().orc_lk_flow_init(prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W, levels, radius, iters, _fp(init), _fp(out))
    if not ok:
        raise ValueError("orc_lk_flow: bad parameters")
    return out


def farneback_layers(W: int, H: int, levels: int = 5):
    """-> [(w_k, h_k) for k = 0 .. K]: the layers calcOpticalFlowFarneback runs (K <= levels; a layer under 32 px ends the pyramid)."""
    K = lib().orc_farneback_layers(W, H, levels)
    out = []
    for k in range(K + 1):
        w, h = C.c_int(0), C.c_int(0)
        lib().orc_farneback_layer_size(W, H, k, C.byref(w), C.byref(h))
        out.append((w.value, h.value))
    return out


def farneback_flow(prev, cur, levels: int = 5, winsize: int = 13, iters: int = 3, poly_n: int = 7, poly_sigma: float = 1.5, init=None) -> np.ndarray:
    """cv-decoder's dense flow (cv-decoder/src/lib.rs:188-199; defaults = its arguments) -> flow[H, W, 2] f32 (dx, dy): prev(x, y) ~
    cur(x + dx, y + dy).  init: None or a [H, W, 2] flow (OPTFLOW_USE_INITIAL_FLOW).  PARITY UNPINNED (OpenCV is not here)."""
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    H, W = prev.shape
    out = np.zeros((H, W, 2), np.float32)
    u8 = C.POINTER(C.c_uint8)
    f = lib().orc_farneback_flow
    f.argtypes = [u8, u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    ini = None if init is None else _f32(init)
    rc = f(prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W, levels, winsize, iters, poly_n, float(poly_sigma),
           None if ini is None else _fp(ini), _fp(out))
    if rc != 0:
        raise ValueError("orc_farneback_flow: bad parameters")
    return out


class farneback_blur_variant:
    """with oracle.farneback_blur_variant(v): ... -- the published form of OpenCV's separable Gaussian the layers are blurred with (0 = the
    SPEC, what every parity test uses; 1 = ascending row taps beyond 5 taps; 2 = 1 with fused multiply-adds).  External kit only."""

    def __init__(self, v: int):
        self.v = int(v)

    def __enter__(self):
        f = lib().orc_farneback_set_blur_variant
        f.argtypes = [C.c_int]; f.restype = C.c_int
        self.prev = f(self.v)
        return self

    def __exit__(self, *a):
        lib().orc_farneback_set_blur_variant(self.prev)


def farneback_layer(img, k: int, poly_n: int = 7, poly_sigma: float = 1.5):
    """Stage-wise view of layer k of one frame: -> (I [h, w] the blurred + resized image, R [h, w, 5] its polynomial expansion)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    w, h = C.c_int(0), C.c_int(0)
    lib().orc_farneback_layer_size(W, H, k, C.byref(w), C.byref(h))
    I = np.zeros((h.value, w.value), np.float32); R = np.zeros((h.value, w.value, 5), np.float32)
    f = lib().orc_farneback_layer_debug
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    assert f(img.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, W, k, poly_n, float(poly_sigma), _fp(I), _fp(R)) == 0
    return I, R


def farneback_kernels(k: int, poly_n: int = 7, poly_sigma: float = 1.5):
    """-> (blur taps of layer k [2 r + 1], g [n + 1], xg, xxg, (ig11, ig03, ig33, ig55))"""
    taps = np.zeros(256, np.float32)
    f = lib().orc_farneback_blur_kernel
    f.argtypes = [C.c_int, C.POINTER(C.c_float)]
    r = f(k, _fp(taps))
    g = np.zeros((3, poly_n + 1), np.float32); ig = (C.c_double * 4)()
    p = lib().orc_farneback_poly_kernel
    p.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double * 4]
    p.restype = None
    p(poly_n, float(poly_sigma), _fp(g[0]), _fp(g[1]), _fp(g[2]), ig)
    return taps[:2 * r + 1].copy(), g[0], g[1], g[2], tuple(ig)


def lk_flow_trace(prev, cur, levels: int = 3, radius: int = 4, iters: int = 3, init=None):
    """-> (flow[H, W, 2], trace): trace[l][it] = the flow [h_l, w_l, 2] entering step `it` of pyramid level l (l = 0 finest)."""
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    H, W = prev.shape
    shapes = [(H, W)]
    for _ in range(1, levels):
        shapes.append(((shapes[-1][0] + 1) // 2, (shapes[-1][1] + 1) // 2))
    out = np.zeros((H, W, 2), np.float32)
    buf = np.zeros(sum(2 * h * w * iters for h, w in shapes), np.float32)
    u8 = C.POINTER(C.c_uint8)
    ini = None if init is None else _f32(init)
    ok = lib().orc_lk_flow_trace(prev.ctypes.data_as(u8), cur.ctypes.data_as(u8), W, H, W, levels, radius, iters,
                                 None if ini is None else _fp(ini), _fp(out), _fp(buf))
    if not ok:
        raise ValueError("orc_lk_flow_trace: bad parameters")
    trace, o = {}, 0
    for l in range(levels - 1, -1, -1):
        h, w = shapes[l]
        trace[l] = buf[o:o + 2 * h * w * iters].reshape(iters, h, w, 2)
        o += 2 * h * w * iters
    return out, trace


def flow_to_entries(flow) -> np.ndarray:
    f = _f32(flow); H, W = f.shape[:2]
    out = np.zeros((H * W, 4), np.float32)
    lib().orc_flow_to_entries(_fp(f), W, H, _fp(out))
    return out


def contrast_mask(gray) -> np.ndarray:
    """cv-decoder's Sobel/threshold/dilate mask (cv-decoder/src/lib.rs:203-237) -> u8[H, W], 1 = keep"""
    g = np.ascontiguousarray(gray, np.uint8); H, W = g.shape
    out = np.zeros((H, W), np.uint8)
    u8 = C.POINTER(C.c_uint8)
    lib().orc_contrast_mask(g.ctypes.data_as(u8), W, H, W, out.ctypes.data_as(u8))
    return out


def masked_flow_to_entries(flow, mask=None) -> np.ndarray:
    f = _f32(flow); H, W = f.shape[:2]
    out = np.zeros((H * W, 4), np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    n = lib().orc_masked_flow_to_entries(_fp(f), None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)), W, H, _fp(out))
    return out[:n].copy()


def sad_simd_level() -> str:
    """Inner loop the timed CPU baseline uses on this host (run-time dispatch)."""
    return {2: "avx2 vmpsadbw (8 candidates per instruction)", 1: "sse2 psadbw", 0: "scalar"}[int(lib().orc_sad_simd_level())]


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> int:
    """Threads of the OpenMP loops that take no thread argument (lk_flow); returns the previous setting."""
    return int(lib().orc_set_num_threads(int(n)))


# ---- cv-decoder's frame front-end (oracle/frontend_oracle.c) ----
FMT_LUMA, FMT_BGR, FMT_RGBA, FMT_BGRA = 0, 1, 2, 3
_FMT_CN = {FMT_LUMA: 1, FMT_BGR: 3, FMT_RGBA: 4, FMT_BGRA: 4}


def cv_grid(W: int, H: int, max_w: int = 150, max_h: int = 150):
    """cv-decoder/src/lib.rs:98-121 -> (gw, gh): the record grid (and, with "Process Fullres" = false, the size the frame is resized to)."""
    gw, gh = C.c_int(0), C.c_int(0)
    lib().orc_cv_grid(int(W), int(H), int(max_w), int(max_h), C.byref(gw), C.byref(gh))
    return gw.value, gh.value


def resize_linear(img, dw: int, dh: int, variant: int = 0) -> np.ndarray:
    """imgproc::resize(.., INTER_LINEAR) of an 8-bit image [H, W] or [H, W, cn] (cv-decoder/src/lib.rs:124-133) -> [dh, dw(, cn)]."""
    a = np.ascontiguousarray(img, np.uint8)
    H, W = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    out = np.zeros((dh, dw) if a.ndim == 2 else (dh, dw, cn), np.uint8)
    u8 = C.POINTER(C.c_uint8)
    f = lib().orc_resize_linear_u8_ex
    f.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_int, u8, C.c_int, C.c_int, C.c_int]
    if f(a.ctypes.data_as(u8), W, H, W * cn, cn, out.ctypes.data_as(u8), int(dw), int(dh), int(variant)) != 0:
        raise ValueError("orc_resize_linear_u8: bad arguments")
    return out


def resize_linear_axis(src: int, dst: int, edge_rule: bool):
    ofs = np.zeros(dst, np.int32); coef = np.zeros((dst, 2), np.int16)
    f = lib().orc_resize_linear_axis
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_short)]
    assert f(src, dst, int(edge_rule), ofs.ctypes.data_as(C.POINTER(C.c_int)), coef.ctypes.data_as(C.POINTER(C.c_short))) == 0
    return ofs, coef


def to_gray(img, fmt: int = FMT_BGR) -> np.ndarray:
    """imgproc::cvt_color(.., COLOR_BGR2GRAY) (cv-decoder/src/lib.rs:135) of [H, W, 3 | 4] u8 -> [H, W] u8; fmt names the byte order."""
    a = np.ascontiguousarray(img, np.uint8)
    H, W, cn = a.shape
    assert cn == _FMT_CN[fmt], (cn, fmt)
    out = np.zeros((H, W), np.uint8)
    u8 = C.POINTER(C.c_uint8)
    f = lib().orc_to_gray_u8
    f.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_int, u8]
    if f(a.ctypes.data_as(u8), W, H, W * cn, int(fmt), out.ctypes.data_as(u8)) != 0:
        raise ValueError("orc_to_gray_u8: bad arguments")
    return out


def cv_frontend(frame, fmt: int = FMT_LUMA, process_fullres: bool = True, max_w: int = 150, max_h: int = 150) -> np.ndarray:
    """What cv-decoder's read loop leaves in `self.gray` for one frame (cv-decoder/src/lib.rs:98-135): [resize to the capped grid] -> gray.
    A luma frame (this build's raw-stream input) has no colour conversion: it is resized as one channel."""
    a = np.ascontiguousarray(frame, np.uint8)
    H, W = a.shape[:2]
    if not process_fullres:
        gw, gh = cv_grid(W, H, max_w, max_h)
        a = resize_linear(a, gw, gh)
    return a if fmt == FMT_LUMA else to_gray(a, fmt)


def cv_decode(prev, cur, fmt: int = FMT_LUMA, process_fullres: bool = True, max_w: int = 150, max_h: int = 150, flow: str = "farneback",
              contrast_mask_on: bool = True, levels=None, radius=None, iters: int = 3, init=None):
    """One cv-decoder process_frame on a pair (cv-decoder/src/lib.rs:82-294) with the stages of this oracle chained:
    front-end -> flow -> contrast mask -> records.  -> (records [n, 4], (grid_w, grid_h), flow [h, w, 2] of the processed frames).
    process_fullres = False: one record per unmasked pixel of the REDUCED frame (:274-276)."""
    H, W = np.asarray(prev).shape[:2]
    g0 = cv_frontend(prev, fmt, process_fullres, max_w, max_h)
    g1 = cv_frontend(cur, fmt, process_fullres, max_w, max_h)
    if flow == "farneback":
        f = farneback_flow(g0, g1, 5 if levels is None else levels, 2 * (6 if radius is None else radius) + 1, iters, init=init)
    else:
        f = lk_flow(g0, g1, 3 if levels is None else levels, 4 if radius is None else radius, iters)
    rec = masked_flow_to_entries(f, contrast_mask(g1) if contrast_mask_on else None)
    gw, gh = cv_grid(W, H, max_w, max_h)
    if process_fullres:
        rec = densify_to_entries(rec, gw, gh)
    return rec, (gw, gh), f
