"""ctypes binding of libofps_hip.so (include/ofps_hip.h).  No fallback: a missing library or a
missing GPU raises -- the product path is the HIP path or nothing."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libofps_hip.so")

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_szp = C.POINTER(C.c_size_t)
_ctx = C.c_void_p
_vp = C.c_void_p

# name -> (restype, argtypes): one row per declaration in include/ofps_hip.h
PROTOTYPES = {
    "ofps_hip_api_version": (C.c_int, []),
    "ofps_hip_device_count": (C.c_int, []),
    "ofps_hip_init": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "ofps_hip_destroy": (None, [_ctx]),
    "ofps_hip_last_error": (C.c_char_p, [_ctx]),
    "ofps_hip_set_stream": (C.c_int, [_ctx, _vp]),
    "ofps_hip_use_own_stream": (C.c_int, [_ctx]),
    "ofps_hip_get_stream": (_vp, [_ctx]),
    "ofps_hip_sync": (C.c_int, [_ctx]),
    "ofps_hip_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_char_p]),
    "ofps_hip_has_test_hooks": (C.c_int, []),
    "ofps_hip_almeida_recoveries": (C.c_int, [_ctx, C.POINTER(C.c_uint64)]),
    "ofps_hip_malloc": (C.c_int, [_ctx, C.c_size_t, C.POINTER(_vp)]),
    "ofps_hip_free": (C.c_int, [_ctx, _vp]),
    "ofps_hip_memcpy_h2d": (C.c_int, [_ctx, _vp, _vp, C.c_size_t]),
    "ofps_hip_memcpy_d2h": (C.c_int, [_ctx, _vp, _vp, C.c_size_t]),
    "ofps_hip_timer_start": (C.c_int, [_ctx]),
    "ofps_hip_timer_stop": (C.c_int, [_ctx, _f32p]),
    "ofps_hip_sad_block_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ofps_hip_set_sad_mode": (C.c_int, [_ctx, C.c_int]),
    "ofps_hip_sad_pruned_overflow_strips": (C.c_int, [_ctx, _u32p]),
    "ofps_hip_sad_flow": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    _f32p, _i32p, _szp]),
    "ofps_hip_sad_flow_dev": (C.c_int, [_ctx, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                        C.c_int, C.c_int, _vp, _vp]),
    "ofps_hip_lk_flow": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p]),
    "ofps_hip_farneback_flow": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, _f32p, _f32p]),
    "ofps_hip_farneback_flow_dev": (C.c_int, [_ctx, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp]),
    "ofps_hip_lk_spec_revision": (C.c_int, []),
    "ofps_hip_lk_helped_tiles": (C.c_int, [_ctx, C.POINTER(C.c_uint64)]),
    "ofps_hip_flow_cache_hits": (C.c_int, [_ctx, C.POINTER(C.c_uint64)]),
    "ofps_hip_frame_channels": (C.c_int, [C.c_int]),
    "ofps_hip_cv_grid": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ofps_hip_resize_linear": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]),
    "ofps_hip_resize_linear_dev": (C.c_int, [_ctx, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    "ofps_hip_cv_frontend": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ofps_hip_cv_frontend_dev": (C.c_int, [_ctx, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ofps_hip_contrast_mask": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _u8p]),
    "ofps_hip_contrast_mask_dev": (C.c_int, [_ctx, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ofps_hip_lk_decode": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_uint, _f32p, _szp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ofps_hip_lk_push_frame": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_uint, _f32p, _szp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ofps_hip_lk_push_frame_async": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_uint, C.POINTER(C.c_int)]),
    "ofps_hip_lk_frame_wait": (C.c_int, [_ctx, C.c_int, _f32p, _szp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ofps_hip_lk_reset": (C.c_int, [_ctx]),
    "ofps_hip_lk_rewind": (C.c_int, [_ctx]),
    "ofps_hip_lk_flow_dev": (C.c_int, [_ctx, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ofps_hip_lk_flow_init_dev": (C.c_int, [_ctx, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "ofps_hip_densify": (C.c_int, [_ctx, _f32p, C.c_size_t, C.c_int, C.c_int, _f32p, _u32p]),
    "ofps_hip_densify_raster_dev": (C.c_int, [_ctx, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int]),
    "ofps_hip_densify_weighted": (C.c_int, [_ctx, _f32p, _f32p, C.c_size_t, C.c_int, C.c_int, _f32p, _u32p]),
    "ofps_hip_densify_dev": (C.c_int, [_ctx, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ofps_hip_densify_to_entries": (C.c_int, [_ctx, _f32p, C.c_size_t, C.c_int, C.c_int, _f32p, _szp]),
    "ofps_hip_densify_interpolated": (C.c_int, [_ctx, _f32p, C.c_size_t, C.c_int, C.c_int, _f32p]),
    "ofps_hip_block_dim": (C.c_int, [C.c_float, C.c_size_t]),
    "ofps_hip_detect": (C.c_int, [_ctx, _f32p, C.c_size_t, C.c_float, C.c_size_t, C.c_float,
                                  C.POINTER(C.c_int), _szp, C.POINTER(C.c_int), _f32p]),
    "ofps_hip_detect_dev": (C.c_int, [_ctx, _vp, C.c_size_t, C.c_int, C.c_float, C.c_size_t, C.c_float, _vp, _vp]),
    "ofps_hip_almeida": (C.c_int, [_ctx, _f32p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_size_t,
                                   C.c_float, C.c_size_t, C.c_uint64, _f32p, _f32p]),
    "ofps_hip_almeida_dev": (C.c_int, [_ctx, _vp, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int, C.c_size_t,
                                       C.c_float, C.c_size_t, C.c_uint64, _vp]),
}



class FrameParams(C.Structure):
    _fields_ = [("block", C.c_int), ("range", C.c_int), ("run_detector", C.c_int), ("min_size", C.c_float),
                ("subdivide", C.c_size_t), ("target_motion", C.c_float), ("run_estimator", C.c_int),
                ("aspect", C.c_float), ("fov_y_deg", C.c_float), ("use_ransac", C.c_int), ("num_iters", C.c_size_t),
                ("inlier_deg", C.c_float), ("num_samples", C.c_size_t), ("seed", C.c_uint64)]


class FrameResult(C.Structure):
    _fields_ = [("have_vectors", C.c_int), ("n_vectors", C.c_size_t), ("has_motion", C.c_int), ("area", C.c_size_t),
                ("dim", C.c_int), ("quat", C.c_float * 4)]


PROTOTYPES["ofps_hip_reset_frames"] = (C.c_int, [_ctx])
PROTOTYPES["ofps_hip_stage_frame"] = (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int])
PROTOTYPES["ofps_hip_host_alloc"] = (C.c_int, [_ctx, C.c_size_t, C.POINTER(C.c_void_p)])
PROTOTYPES["ofps_hip_host_free"] = (C.c_int, [_ctx, C.c_void_p])
PROTOTYPES["ofps_hip_push_frame"] = (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(FrameParams),
                                               C.POINTER(FrameResult), _f32p, _f32p])

PROTOTYPES["ofps_hip_push_frame_async"] = (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(FrameParams), _f32p, _f32p,
                                                     C.POINTER(C.c_int)])
PROTOTYPES["ofps_hip_frame_wait"] = (C.c_int, [_ctx, C.c_int, C.POINTER(FrameResult)])
PROTOTYPES["ofps_hip_push_frames_async"] = (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(FrameParams), _f32p,
                                                      C.POINTER(C.c_int)])
PROTOTYPES["ofps_hip_frames_wait"] = (C.c_int, [_ctx, C.c_int, C.POINTER(FrameResult)])

_multi = C.c_void_p
PROTOTYPES["ofps_hip_checksum_dev"] = (C.c_int, [_ctx, _vp, C.c_size_t, C.c_int, _vp])
PROTOTYPES["ofps_hip_multi_init"] = (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_multi)])
PROTOTYPES["ofps_hip_multi_destroy"] = (None, [_multi])
PROTOTYPES["ofps_hip_multi_last_error"] = (C.c_char_p, [_multi])
PROTOTYPES["ofps_hip_multi_worker_count"] = (C.c_int, [_multi])
PROTOTYPES["ofps_hip_multi_fanout"] = (C.c_int, [_multi, C.POINTER(C.c_uint64)])
PROTOTYPES["ofps_hip_multi_pair_range"] = (None, [C.c_size_t, C.c_int, C.c_int, _szp, _szp])
PROTOTYPES["ofps_hip_multi_frame_range"] = (None, [C.c_size_t, C.c_int, C.c_int, C.c_int, _szp, _szp])
PROTOTYPES["ofps_hip_multi_sad_flow"] = (C.c_int, [_multi, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, _f32p])
PROTOTYPES["ofps_hip_multi_stage_frames"] = (C.c_int, [_multi, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int])
PROTOTYPES["ofps_hip_multi_run_resident"] = (C.c_int, [_multi, C.c_int, C.c_int, C.c_int, _f32p])
PROTOTYPES["ofps_hip_multi_fetch"] = (C.c_int, [_multi, C.c_int, _f32p])
PROTOTYPES["ofps_hip_multi_push_frames_async"] = (C.c_int, [_multi, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(FrameParams), _f32p,
                                                            C.POINTER(C.c_int)])
PROTOTYPES["ofps_hip_multi_frames_wait"] = (C.c_int, [_multi, C.c_int, C.POINTER(FrameResult)])
PROTOTYPES["ofps_hip_multi_reset_frames"] = (C.c_int, [_multi])
PROTOTYPES["ofps_hip_multi_stream_plan"] = (None, [C.c_long, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)])

_lib = None
_lib_hooks = None
LIB_HOOKS_PATH = os.path.join(HERE, "libofps_hip_testhooks.so")


class OfpsHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libofps_hip error {code}: {message}")
        self.code = code


def _open(path: str):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m ofps_amd.build` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (plumbing only: shares the HIP runtime, streams, distributed)
    except Exception:
        pass
    lib = C.CDLL(path)
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError(f"{path} does not export: {', '.join(missing)}")
    return lib


def load():
    """dlopen libofps_hip.so.  torch (when importable) is imported first so that both share one
    libamdhip64.so.7 (torch bundles its own copy; the SONAMEs match, the first one loaded wins)."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def load_test_hooks():
    """The same library built with -DOFPS_HIP_TEST_HOOKS: the only build whose fault injectors can be armed
    (ofps_hip_set_option).  Parity tests only; nothing in the product package loads it."""
    global _lib_hooks
    if _lib_hooks is None:
        _lib_hooks = _open(LIB_HOOKS_PATH)
        assert _lib_hooks.ofps_hip_has_test_hooks() == 1
    return _lib_hooks
