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
    assert lib.ofps_hip_api_version() == 2
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


# ---- the three statements of the ABI agree argument by argument (loader: ofps/src/plugins/mod.rs:35,139-160)

def test_ctypes_table_matches_the_header_argument_by_argument():
    import ffi_parse as F
    hdr = F.header_signatures()
    assert sorted(hdr) == _declared_symbols()
    for name, (res, args) in _lib.PROTOTYPES.items():
        h_ret, h_args = hdr[name]
        assert len(h_args) == len(args), f"{name}: header has {len(h_args)} arguments, ctypes {len(args)}"
        assert F.same_class(h_ret, F.ctypes_class(res), lp64=True), f"{name}: return {h_ret} vs {F.ctypes_class(res)}"
        for i, (a, b) in enumerate(zip(h_args, args)):
            assert F.same_class(a, F.ctypes_class(b), lp64=True), f"{name}: argument {i}: header {a}, ctypes {F.ctypes_class(b)}"


def test_integration_md_rust_binding_matches_the_header():
    """INTEGRATION.md's ffi.rs is source only (no rustc here): at least its extern "C" block must say what the header says --
    name, arity, pointer / integer / float class, integer width, constness and pointee of every argument and the return."""
    import ffi_parse as F
    hdr, rs = F.header_signatures(), F.rust_signatures()
    assert len(rs) >= 36
    for name, (ret, args) in rs.items():
        assert name in hdr, f"{name} bound in INTEGRATION.md but not declared in include/ofps_hip.h"
        h_ret, h_args = hdr[name]
        assert len(h_args) == len(args), f"{name}: header has {len(h_args)} arguments, the Rust binding {len(args)}"
        assert F.same_class(h_ret, ret), f"{name}: return {h_ret} vs {ret}"
        for i, (a, b) in enumerate(zip(h_args, args)):
            assert F.same_class(a, b), f"{name}: argument {i}: header {a}, Rust {b}"
    # every entry point the shim's text calls is declared in the block
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    used = set(re.findall(r"\b(ofps_hip_[a-z0-9_]+)\s*\(", re.sub(r"`[^`\n]*`", "", text)))
    assert not (used - set(rs) - {"ofps_hip_ctx", "ofps_hip_multi"}), sorted(used - set(rs))
    # what the loader checks and what the Python plugin exposes: the version gate and the pruning switch are bound AND used
    assert "ofps_hip_api_version" in rs and "ofps_hip_set_sad_mode" in rs
    assert re.search(r"let v = unsafe \{ ofps_hip_api_version\(\) \};\s*if v != OFPS_HIP_API_VERSION", text), "Ctx::new does not check the API version"
    assert '"Exact pruning"' in text                                 # the property name plugins.py / ofps_host.cpp expose


def test_frame_structs_agree_field_by_field():
    import ffi_parse as F
    for c_name, rs_name, ct in (("ofps_hip_frame_params", "FrameParams", _lib.FrameParams),
                                ("ofps_hip_frame_result", "FrameResult", _lib.FrameResult)):
        h = F.header_struct_fields(c_name)
        assert h == F.rust_struct_fields(rs_name), c_name
        got = []
        for fname, ftype in ct._fields_:
            if hasattr(ftype, "_length_"):
                got.append((fname, F.ctypes_class(ftype._type_), ftype._length_))
            else:
                got.append((fname, F.ctypes_class(ftype), 0))
        assert [f[0] for f in got] == [f[0] for f in h], c_name
        for (n, a, la), (_, b, lb) in zip(h, got):
            assert la == lb and F.same_class(a, b, lp64=True), (c_name, n)


def test_committed_lk_rows9_is_what_the_generator_writes(tmp_path):
    """5.7k lines of generated inline asm whose wait counts the generator derives: a stale or hand-edited copy would only
    show up in the GPU bit-exactness tests."""
    import subprocess
    import sys
    out = tmp_path / "lk_rows9.inc"
    gen = os.path.join(ROOT, "tools", "gen_lk_rows9.py")
    p = subprocess.run([sys.executable, gen, str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    if not out.exists():                                             # the generator writes to stdout
        out.write_text(p.stdout)
    assert out.read_bytes() == open(os.path.join(ROOT, "ofps_amd", "csrc", "lk_rows9.inc"), "rb").read()
