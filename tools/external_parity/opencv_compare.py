#!/usr/bin/env python3
"""External check of the hip_flow decoder's algorithm against the REAL thing the reference calls (anyone with OpenCV can run it; this
build's image has no cv2, so the result is not part of the build's own evidence):

    cv-decoder/src/lib.rs:188-199   calc_optical_flow_farneback(old_gray, gray, flow, 0.5, 5, 13, 3, 7, 1.5, flags)

data/farneback_pairs.npz holds two seeded 256 x 144 luma pairs (a smooth camera rotation; region-wise integer motion with flow
discontinuities) and the flow this build's CPU restatement (oracle/farneback_oracle.c, bit-identical to the HIP kernels:
tests/test_farneback_gpu.py) computes for them, cold and started from that flow (make_farneback_pairs.py made the second set).  The script runs cv2.calcOpticalFlowFarneback with the reference's arguments on the
same frames and prints how far apart the two flows are.

    python opencv_compare.py [data/farneback_pairs.npz]            # needs numpy + opencv-python, nothing from this repository

What to expect: every stage is restated with the precision OpenCV's CPU path uses, but OpenCV's SIMD builds fuse and reorder
float operations, its resize / Gaussian kernels differ in the last bit, and Farneback's 2 x 2 solve amplifies that where the window has
no texture -- so not bit-equal; a median difference around 1e-4 px and a 99th percentile under 1e-2 px say "same algorithm"."""
import sys

import numpy as np


def main():
    import cv2
    path = sys.argv[1] if len(sys.argv) > 1 else __file__.rsplit("/", 1)[0] + "/data/farneback_pairs.npz"
    d = np.load(path)
    ok = True
    print("OpenCV", cv2.__version__)
    for name in ("camera", "regions"):
        prev, cur, ours = d[name + "_prev"], d[name + "_cur"], d[name + "_flow"]
        cv = cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 5, 13, 3, 7, 1.5, 0)
        diff = np.linalg.norm(cv - ours, axis=2)
        mag = np.linalg.norm(cv, axis=2)
        q = np.percentile(diff, [50, 90, 99, 100])
        print(f"{name:8s} {prev.shape[1]}x{prev.shape[0]}  |flow| mean {mag.mean():.3f} px   |cv2 - build| median {q[0]:.2e}  p90 {q[1]:.2e}  p99 {q[2]:.2e}  "
              f"max {q[3]:.2e} px")
        ok = ok and q[0] < 2e-3 and q[2] < 5e-2
        # the same pair started from that flow (OPTFLOW_USE_INITIAL_FLOW, what cv-decoder passes from its second frame on: cv-decoder/src/
        # lib.rs:161-165): OpenCV area-resizes the flow to the coarsest layer and scales it; both sides start from the BUILD's cold flow
        if name + "_flow_warm" in d:
            cvw = cv2.calcOpticalFlowFarneback(prev, cur, ours.copy(), 0.5, 5, 13, 3, 7, 1.5, cv2.OPTFLOW_USE_INITIAL_FLOW)
            qw = np.percentile(np.linalg.norm(cvw - d[name + "_flow_warm"], axis=2), [50, 90, 99, 100])
            print(f"{'':8s} with OPTFLOW_USE_INITIAL_FLOW               |cv2 - build| median {qw[0]:.2e}  p90 {qw[1]:.2e}  p99 {qw[2]:.2e}  max {qw[3]:.2e} px")
            ok = ok and qw[0] < 2e-3 and qw[2] < 5e-2
    print("verdict:", "same algorithm (within the stated bounds)" if ok else "DIFFERENT -- please report the numbers above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
