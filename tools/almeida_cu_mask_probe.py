#!/usr/bin/env python3
"""The small cluster Almeida solve on CU-masked streams: recoveries and time per call, flat vs one-XCD exchange."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext
hip = C.CDLL("libamdhip64.so")
ctx = HipContext(0)
e = synth.rotation_field(120, 67)
q_ref, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
ctx.set_option("OFPS_HIP_ALMEIDA_PATH", "cluster")
for mask in ("none", "first32", "first64", "every_other", "every_eighth", "every_eighth_x2"):
    cus = {"none": range(256), "first32": range(32), "first64": range(64), "every_other": range(0, 256, 2), "every_eighth": range(0, 256, 8),
           "every_eighth_x2": [c for c in range(256) if c % 8 < 2]}[mask]
    words = (C.c_uint32 * 8)(*([0] * 8))
    for cu in cus: words[cu // 32] |= 1 << (cu % 32)
    stream = C.c_void_p(0)
    assert hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, words) == 0
    ctx.set_stream(stream.value)
    for mode in (0, 1):
        ctx.set_option("OFPS_HIP_ALMEIDA_ONE_XCD", mode)
        r0 = ctx.almeida_recoveries()
        t0 = time.perf_counter()
        for _ in range(5): q, _ = ctx.almeida(e, 16 / 9, 22.275, use_ransac=False)
        dt = (time.perf_counter() - t0) / 5 * 1e3
        print(f"mask {mask:16s} one_xcd={mode}: {dt:8.3f} ms per call, recoveries {ctx.almeida_recoveries() - r0} of 5, same bits {bool((q.view(np.uint32) == q_ref.view(np.uint32)).all())}", flush=True)
    ctx.use_own_stream(); hip.hipStreamDestroy(stream)
