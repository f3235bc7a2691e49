import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle, indep_model as im, almeida_cases as ac
from ofps_amd.runtime import HipContext
ctx = HipContext(0)
worst = 0.0
cam = oracle.camera(1.0, 90.0)
for rot in (0.1, 1.0, 10.0):
    for (r, p, y) in ac.angle_combos(rot)[1:]:
        q_i, ent, keep = im.almeida_test_field(r, p, y, n=300)
        e = ent[keep].astype(np.float32)
        est, _ = ctx.almeida(e, 1.0, 90.0, use_ransac=False)
        dev = np.abs(est - oracle.solve_ypr_given(e, cam)).max()
        worst = max(worst, dev)
print("dense regime, reference camera, rotations to 10 deg: worst |q_hip - q_oracle| = %.3g (bound 2e-6)" % worst)
from ofps_amd import synth
d = synth.rotation_field(1920, 1080)
est, _ = ctx.almeida(d, 16 / 9, 22.275, use_ransac=False)
print("1080p per-pixel field: |dq| = %.3g" % np.abs(est - oracle.solve_ypr_given(d, oracle.camera(16 / 9, 22.275))).max())
