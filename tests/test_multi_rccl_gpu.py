"""The optional RCCL fan-out of the shared key frame inside ofps_hip_multi_* (OFPS_HIP_MULTI_RCCL=1: one ncclBroadcast over a
communicator of the workers' devices instead of hipMemcpyPeerAsync; north_star: "RCCL broadcast of shared reference frames over xGMI").
One GPU per box here: a communicator of ONE rank, whose broadcast is a self-copy -- the whole code path (dlopen of librccl.so,
ncclCommInitAll, group call, per-worker stream, synchronisation) runs; the vectors must equal the copy path's and the oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from ofps_amd import _lib, synth
import oracle
lib = _lib.load()
W, H, B, R, F = 640, 360, 16, 16, 5
fr = synth.luma_sequence(F, W, H, max_step=8, seed=12)
devs = (C.c_int * 1)(0)
m = C.c_void_p(0)
assert lib.ofps_hip_multi_init(devs, 1, C.byref(m)) == 0, lib.ofps_hip_multi_last_error(None)
nb = lib.ofps_hip_sad_block_count(W, H, B)
out = np.zeros((F - 1, nb, 4), np.float32)
rc = lib.ofps_hip_multi_sad_flow(m, fr.ctypes.data_as(C.POINTER(C.c_uint8)), F, W, H, W, W * H, 1, B, R, out.ctypes.data_as(C.POINTER(C.c_float)))
assert rc == 0, lib.ofps_hip_multi_last_error(m)
n = C.c_uint64(0)
mode = lib.ofps_hip_multi_fanout(m, C.byref(n))
ok = all(np.array_equal(out[k].view(np.uint32), oracle.sad_flow(fr[0], fr[k + 1], B, R)[0].view(np.uint32)) for k in range(F - 1))
lib.ofps_hip_multi_destroy(m)
print(json.dumps({"fanout": mode, "broadcasts": n.value, "vectors_equal_the_oracle_key_frame_pairs": bool(ok)}))
"""


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])      # (RCCL prints its version banner to stdout at exit)


def test_key_frame_fan_out_through_rccl_world_size_one():
    plain = _run({"OFPS_HIP_MULTI_RCCL": "0"})
    assert plain == {"fanout": 0, "broadcasts": 0, "vectors_equal_the_oracle_key_frame_pairs": True}
    rccl = _run({"OFPS_HIP_MULTI_RCCL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert rccl["vectors_equal_the_oracle_key_frame_pairs"] is True
    assert rccl["fanout"] == 1 and rccl["broadcasts"] == 1, rccl          # librccl.so is part of the image: the communicator must come up
