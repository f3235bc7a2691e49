"""Builds libofps_hip.so (the C-ABI product library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.  The
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libofps_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar f32 adds/muls into v_pk_add_f32 / v_pk_mul_f32, which issue
# at half the rate of the scalar forms on gfx950 (profiles/ubench_valu_r02.txt: 4.8 vs 2.5 cycles per wave64
# instruction) -- no flops gained -- and need their operands in aligned register pairs: 31 extra v_mov per window row in
# the LK step kernel.  Per-lane results are identical either way (IEEE f32).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"] + os.environ.get("OFPS_HIP_EXTRA_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


# The parity tests' fault injectors (a withheld granule in the cluster Almeida solver, a forced grouped path in the LK level
# kernel) are compiled only into this second library; the product library refuses to arm them.
LIB_HOOKS = os.path.join(HERE, "libofps_hip_testhooks.so")
HOOKED = ("ctx", "almeida", "lk")            # translation units that look at OFPS_HIP_TEST_HOOKS


def _compile(src: str, obj: str, extra, force: bool, verbose: bool, hdrs) -> None:
    if force or _stale(obj, [src] + hdrs):
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)


def _link(lib: str, objs, force: bool, verbose: bool) -> None:
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = False) -> str:
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objs, objs_hooks = [], []
    for src in sources():
        stem = os.path.basename(src)[:-4]
        obj = os.path.join(CSRC, stem + ".o")
        _compile(src, obj, [], force, verbose, hdrs)
        objs.append(obj)
        if stem in HOOKED:
            obj_h = os.path.join(CSRC, stem + ".hooks.o")
            _compile(src, obj_h, ["-DOFPS_HIP_TEST_HOOKS"], force, verbose, hdrs)
            objs_hooks.append(obj_h)
        else:
            objs_hooks.append(obj)
    _link(LIB, objs, force, verbose)
    _link(LIB_HOOKS, objs_hooks, force, verbose)
    return LIB


HOST = os.path.join(HERE, "host")
TOOL = os.path.join(HOST, "ofps_hip_tool")


def build_host(force: bool = False, verbose: bool = False) -> str:
    """C++ host layer + CLI (g++; links libofps_hip.so, no device code)."""
    srcs = [os.path.join(HOST, "ofps_host.cpp"), os.path.join(HOST, "ofps_hip_tool.cpp")]
    deps = srcs + [os.path.join(HOST, "ofps_host.hpp"), LIB]
    if force or _stale(TOOL, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", TOOL] + srcs + \
              ["-L" + HERE, "-lofps_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return TOOL


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
