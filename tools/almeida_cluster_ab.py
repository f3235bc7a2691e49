#!/usr/bin/env python3
"""A/B of the LSQ drivers on one MI355X: cluster solver (one launch, granule exchange) vs one launch per step vs the
one-workgroup kernel, over problem sizes; prints JSON with ms per estimate and the quaternion each path returns.
Paths are forced with the OFPS_HIP_ALMEIDA_PATH / OFPS_HIP_ALMEIDA_EPT switches (ofps_hip_set_option on the live context)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext


def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


ctx = HipContext(0); ctx.use_torch_stream()


def setenv(path, ept):
    for k, v in (("OFPS_HIP_ALMEIDA_PATH", path), ("OFPS_HIP_ALMEIDA_EPT", ept)):
        ctx.set_option(k, v)



out = {}
sizes = [(64, 36), (120, 67), (150, 84), (240, 135), (480, 270), (960, 540), (1920, 1080)]
for (w, h) in sizes:
    n = w * h
    e = synth.rotation_field(w, h)
    d = torch.from_numpy(e).cuda()
    q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
    row = {}
    variants = [("default", None, None), ("step", "step", None)]
    if n <= 8192: variants += [("wg", "wg", None)]
    if n <= 65536:
        variants += [(f"cluster_ept{k}", "cluster", k) for k in (1, 2, 4, 8) if (n + k * 1024 - 1) // (k * 1024) <= 256]
    for name, path, ept in variants:
        setenv(path, ept)
        ms = timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr()))
        row[name] = {"ms": round(ms, 4), "q": [float(x) for x in q.cpu().numpy()[0]]}
    setenv(None, None)
    ref = np.array(row["step"]["q"])
    for name in row:
        row[name]["max_abs_diff_vs_step"] = float(np.abs(np.array(row[name]["q"]) - ref).max())
        del row[name]["q"]
    out[f"n{n}"] = row
# batches through the cluster path: cfg4-sized vector sets (129,600 per pair)
for (w, h, batch) in ((480, 270, 8), (480, 270, 64), (150, 84, 16)):
    n = w * h
    e = np.stack([synth.rotation_field(w, h, seed=k) for k in range(4)])
    e = np.concatenate([e] * ((batch + 3) // 4))[:batch]
    d = torch.from_numpy(np.ascontiguousarray(e)).cuda()
    q = torch.empty((batch, 4), dtype=torch.float32, device="cuda")
    row = {}
    for name, path in (("default", None), ("step", "step")):
        setenv(path, None)
        ms = timeit(lambda: ctx.almeida_dev(d.data_ptr(), n, batch, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr()), n=10)
        row[name] = {"ms": round(ms, 4), "q": q.cpu().numpy().copy()}
    setenv(None, None)
    diff = float(np.abs(row["default"]["q"] - row["step"]["q"]).max())
    out[f"n{n}_b{batch}"] = {"default_ms": row["default"]["ms"], "step_ms": row["step"]["ms"], "max_abs_diff": diff}
print(json.dumps(out, indent=1))
