import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle, indep_model as im, almeida_cases as ac
from ofps_amd import synth
from ofps_amd.runtime import HipContext
ctx = HipContext(0); ctx.use_torch_stream()
for mode in ("0", "1"):
    ctx.set_option("OFPS_HIP_ALMEIDA_FAST", mode)
    worst = 0.0
    for (w, h) in ((120, 67), (150, 84), (240, 135)):
        d = synth.rotation_field(w, h); n = w * h
        dd = torch.from_numpy(d).cuda(); q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
        f = lambda: ctx.almeida_dev(dd.data_ptr(), n, 1, 16 / 9, 22.275, False, 0, 0.05, 0, 0, q.data_ptr())
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
        dev = np.abs(q.cpu().numpy().ravel() - oracle.solve_ypr_given(d, oracle.camera(16 / 9, 22.275))).max()
        print(f"FAST={mode} n={n}: {ms:.4f} ms  |dq| = {dev:.3g}")
    cam = oracle.camera(1.0, 90.0)
    for rot in (0.1, 1.0, 10.0):
        for (r, p, y) in ac.angle_combos(rot)[1:]:
            q_i, ent, keep = im.almeida_test_field(r, p, y, n=120)
            e = ent[keep].astype(np.float32)
            if len(e) <= 4096: continue
            de = torch.from_numpy(e).cuda(); q = torch.empty((1, 4), dtype=torch.float32, device="cuda")
            ctx.almeida_dev(de.data_ptr(), len(e), 1, 1.0, 90.0, False, 0, 0.05, 0, 0, q.data_ptr()); torch.cuda.synchronize()
            worst = max(worst, np.abs(q.cpu().numpy().ravel() - oracle.solve_ypr_given(e, cam)).max())
    print(f"FAST={mode}: reference camera, ~11k vectors, rotations to 10 deg: worst |dq| = {worst:.3g}")
