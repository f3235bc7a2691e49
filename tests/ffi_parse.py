"""Parsers for the three statements of the C ABI -- include/ofps_hip.h (C prototypes), INTEGRATION.md's `extern "C"` block (the
Rust binding a maintainer adds; ofps/src/plugins/mod.rs:35,139-160 is the loader it plugs into) and ofps_amd/_lib.py (ctypes) --
reduced to one comparable signature form: a tuple of argument classes + a return class, where a class is
    ("ptr", const?, pointee)   pointee in {"u8","i32","u32","u64","f32","usize","char","void","ctx","multi","params","result","ptr"}
    "i32" | "u32" | "i64" | "u64" | "usize" | "f32" | "void"
Test infrastructure only."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_C_SCALAR = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "uint32_t": "u32", "int32_t": "i32", "long": "i64",
             "uint64_t": "u64", "size_t": "usize", "float": "f32", "void": "void", "uint8_t": "u8", "char": "char",
             "ofps_hip_ctx": "ctx", "ofps_hip_multi": "multi", "ofps_hip_frame_params": "params",
             "ofps_hip_frame_result": "result"}
_RS_SCALAR = {"c_int": "i32", "i32": "i32", "u32": "u32", "c_uint": "u32", "c_long": "i64", "i64": "i64", "u64": "u64",
              "usize": "usize", "f32": "f32", "u8": "u8", "c_char": "char", "c_void": "void", "ofps_hip_ctx": "ctx",
              "ofps_hip_multi": "multi", "FrameParams": "params", "FrameResult": "result"}


def _strip_c_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def _c_type(t):
    t = " ".join(t.split())
    stars = t.count("*")
    base = t.replace("*", " ")
    const = bool(re.search(r"\bconst\b", base))
    base = " ".join(w for w in base.split() if w != "const")
    if base not in _C_SCALAR:
        raise ValueError(f"unknown C type {t!r}")
    cls = _C_SCALAR[base]
    if stars == 0:
        return cls
    if stars == 2:
        return ("ptr", False, "ptr:" + cls)
    return ("ptr", const, cls)


def header_signatures(path=None):
    """-> {name: (ret, [arg classes])} for every ofps_hip_* prototype of include/ofps_hip.h"""
    text = _strip_c_comments(open(path or os.path.join(ROOT, "include", "ofps_hip.h")).read())
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ofps_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        ret = ret.split("\n")[-1].strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                a = re.sub(r"\[\d*\]$", "*", a)                          # float out_quat[4] -> pointer
                # the parameter name is the last identifier unless the declaration is a bare type
                parts = re.match(r"^(.*[\s\*])([A-Za-z_]\w*)(\*?)$", a)
                typ = (parts.group(1) + parts.group(3)) if parts else a
                alist.append(_c_type(typ))
        out[name] = (_c_type(ret), alist)
    return out


def header_struct_fields(name, path=None):
    """-> [(field, class, array_len)] of `typedef struct { ... } name;`"""
    text = _strip_c_comments(open(path or os.path.join(ROOT, "include", "ofps_hip.h")).read())
    m = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + name + r"\s*;", text)
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        first = re.match(r"^([A-Za-z_]\w*(?:\s+[A-Za-z_]\w*)*?)\s+(.+)$", decl)
        typ, names = first.group(1), first.group(2)
        for n in names.split(","):
            n = n.strip()
            arr = re.match(r"^(\w+)\[(\d+)\]$", n)
            fields.append((arr.group(1), _C_SCALAR[typ], int(arr.group(2))) if arr else (n, _C_SCALAR[typ], 0))
    return fields


def _rs_type(t):
    t = " ".join(t.split())
    m = re.match(r"^\*(mut|const)\s+(.*)$", t)
    if not m:
        if t not in _RS_SCALAR:
            raise ValueError(f"unknown Rust type {t!r}")
        return _RS_SCALAR[t]
    inner = m.group(2)
    m2 = re.match(r"^\*(mut|const)\s+(.*)$", inner)
    if m2:
        return ("ptr", False, "ptr:" + _RS_SCALAR[m2.group(2)])
    return ("ptr", m.group(1) == "const", _RS_SCALAR[inner])


def rust_ffi_block(path=None):
    text = open(path or os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index('extern "C" {')
    depth, i = 0, start
    while True:
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    return text[start:i + 1]


def rust_signatures(path=None):
    """-> {name: (ret, [arg classes])} for every `pub fn` of INTEGRATION.md's extern "C" block"""
    block = _strip_c_comments(rust_ffi_block(path))
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(ofps_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "").strip()
        alist = []
        if args:
            for a in args.split(","):
                a = a.strip()
                if a:
                    alist.append(_rs_type(a.split(":", 1)[1]))
        out[name] = (_rs_type(ret) if ret else "void", alist)
    return out


def rust_struct_fields(name, path=None):
    text = open(path or os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"pub struct " + name + r"\s*\{([^}]*)\}", text)
    fields = []
    for f in m.group(1).split(","):
        f = " ".join(f.split())
        if not f:
            continue
        n, t = f.replace("pub ", "").split(":")
        arr = re.match(r"^\[(\w+);\s*(\d+)\]$", t.strip())
        fields.append((n.strip(), _RS_SCALAR[arr.group(1)], int(arr.group(2))) if arr else (n.strip(), _RS_SCALAR[t.strip()], 0))
    return fields


_CT_SCALAR = {C.c_int: "i32", C.c_int32: "i32", C.c_uint: "u32", C.c_uint32: "u32", C.c_long: "i64", C.c_uint64: "u64",
              C.c_size_t: "usize", C.c_float: "f32", None: "void"}
_CT_POINTEE = {C.c_uint8: "u8", C.c_float: "f32", C.c_int: "i32", C.c_int32: "i32", C.c_uint32: "u32", C.c_uint: "u32",
               C.c_uint64: "u64", C.c_size_t: "usize", C.c_void_p: "ptr"}


def ctypes_class(t):
    """ctypes type -> comparable class; pointers lose const (ctypes has none) and opaque handles are c_void_p"""
    if t in _CT_SCALAR and t is not C.c_void_p:
        return _CT_SCALAR[t]
    if t is C.c_void_p or t is C.c_char_p:
        return ("ptr", None, "char" if t is C.c_char_p else "any")
    if hasattr(t, "_type_") and not isinstance(t._type_, str):          # POINTER(x)
        inner = t._type_
        if inner in _CT_POINTEE:
            return ("ptr", None, _CT_POINTEE[inner])
        return ("ptr", None, {"FrameParams": "params", "FrameResult": "result"}.get(inner.__name__, inner.__name__))
    raise ValueError(f"unknown ctypes type {t!r}")


def same_class(c_cls, other, ignore_const=False, lp64=False):
    """header class vs a binding's class.  ctypes (lp64=True): c_void_p stands for any pointer, and c_uint64 IS c_size_t
    (both c_ulong on LP64), so the two widths cannot be told apart there."""
    def eq(a, b):
        return a == b or (lp64 and {a, b} == {"u64", "usize"})
    if isinstance(c_cls, tuple) != isinstance(other, tuple):
        return False
    if not isinstance(c_cls, tuple):
        return eq(c_cls, other)
    _, cconst, cpointee = c_cls
    _, oconst, opointee = other
    if opointee == "any":
        return True
    if opointee == "ptr" and cpointee.startswith("ptr:"):
        return True
    if not eq(cpointee, opointee):
        return False
    return ignore_const or oconst is None or cconst == oconst
