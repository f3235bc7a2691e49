#!/usr/bin/env python3
"""Adds the OPTFLOW_USE_INITIAL_FLOW case to data/farneback_pairs.npz (run in the build's container; writes inputs and EXPECTED outputs
only): for each committed pair, the flow this build's CPU restatement (oracle/farneback_oracle.c) computes when it starts from the
pair's own cold-start flow -- what cv-decoder does from its second frame on (cv-decoder/src/lib.rs:161-165).  The frames and the
cold-start flows already in the file are left as they are (tests/test_farneback_oracle.py keeps every array equal to the oracle)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

path = os.path.join(ROOT, "tools", "external_parity", "data", "farneback_pairs.npz")
d = dict(np.load(path))
for name in ("camera", "regions"):
    cold = oracle.farneback_flow(d[name + "_prev"], d[name + "_cur"])
    assert np.array_equal(cold.view(np.uint32), d[name + "_flow"].view(np.uint32)), "the committed cold-start flow is no longer the oracle's"
    d[name + "_flow_warm"] = oracle.farneback_flow(d[name + "_prev"], d[name + "_cur"], init=cold)
np.savez_compressed(path, **d)
print({k: v.shape for k, v in d.items()})
