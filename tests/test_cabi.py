"""CPU tests of the drop-in boundary: libofps_hip.so loads, exports every symbol include/ofps_hip.h
declares, and refuses to run without a GPU (no compute calls are made here)."""
import ctypes as C
import os
import re

import pytest

from ofps_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ofps_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ofps_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m ofps_amd.build` (or __graft_entry__.build())"


def test_every_declared_symbol_is_exported_and_bound():
    lib = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ofps_hip.h but not exported"
    # the ctypes prototype table covers exactly the header
    assert sorted(_lib.PROTOTYPES) == declared


def test_api_version_and_pure_helpers():
    lib = _lib.load()
    assert lib.ofps_hip_api_version() == 1
    assert lib.ofps_hip_sad_block_count(1920, 1080, 16) == 120 * 67          # full blocks only
    assert lib.ofps_hip_sad_block_count(640, 360, 16) == 40 * 22
    assert lib.ofps_hip_sad_block_count(3840, 2160, 8) == 480 * 270
    assert lib.ofps_hip_block_dim(0.05, 3) == 14                              # detector defaults
    assert lib.ofps_hip_block_dim(0.01, 16) == 160


def test_fault_injectors_live_only_in_the_test_hooks_library():
    """libofps_hip.so is built without OFPS_HIP_TEST_HOOKS; libofps_hip_testhooks.so (same sources, same ABI) with it.
    The product library never reads the environment outside ofps_hip_init and holds no injector code path."""
    assert _lib.load().ofps_hip_has_test_hooks() == 0
    assert _lib.load_test_hooks().ofps_hip_has_test_hooks() == 1
    for name in _declared_symbols():
        assert hasattr(_lib.load_test_hooks(), name)
    csrc = os.path.join(ROOT, "ofps_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            src = open(os.path.join(csrc, f)).read()
            if f != "ctx.hip":
                assert "getenv" not in src, f"{f} reads the environment on a call path"
    ctx_src = open(os.path.join(csrc, "ctx.hip")).read()
    assert ctx_src.count("getenv(") == 1                                     # the one loop in ofps_hip_init
    # nothing in the product package binds the hooks library except the explicit test_hooks=True switch of the runtime
    for f in ("plugins.py", "distributed.py", "mvec.py", "synth.py", "build.py"):
        assert "load_test_hooks" not in open(os.path.join(ROOT, "ofps_amd", f)).read()
    assert "load_test_hooks" not in open(os.path.join(ROOT, "bench.py")).read()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ofps_amd.runtime import HipContext, OfpsHipError
    with pytest.raises(OfpsHipError) as ei:
        HipContext(0)
    assert "no CPU fallback" in str(ei.value)


def test_product_package_never_imports_the_oracle():
    """Static check of rule (3): nothing under ofps_amd/ may import or link oracle/."""
    pkg = os.path.join(ROOT, "ofps_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "ofps_oracle.h" not in src and "libofps_oracle" not in src, f
