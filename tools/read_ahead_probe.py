#!/usr/bin/env python3
"""Where the per-frame read-ahead time goes, by library variant (end_to_end.read_ahead went from 0.059 ms per
frame in round 3 to 0.17-0.28 in all three round-4 runs).  Raw ctypes on the five entry points the loop uses, so the same script
drives this tree's library, its A/B builds (tools/read_ahead_bisect.sh) and the round-3 library.

  python tools/read_ahead_probe.py --lib <libofps_hip.so> [--no-torch] [--reps 7]     one configuration, JSON line
  python tools/read_ahead_probe.py --all                                              every variant, each in a fresh process

Per configuration: the three loops of bench.py's end_to_end leg (sync, read-ahead, read-ahead + host copy), `reps` repeats
each, interleaved; per loop the split push / wait / other and the slowest single wait; the cgroup's CPU-throttling counters
around the lot (a throttled process looks like a slow loop)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def throttle():
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(ln.split() for ln in open(p).read().splitlines())
            return {k: int(d[k]) for k in ("nr_periods", "nr_throttled") if k in d} | \
                   {"throttled_us": int(d.get("throttled_usec", d.get("throttled_time", 0)))}
        except OSError:
            continue
    return {}


def one(args):
    import numpy as np
    if args.no_torch:
        sys.modules["torch"] = None                     # `import torch` raises ImportError
    else:
        import torch                                    # noqa: F401  (what ofps_amd._lib does before dlopen)
    from ofps_amd._lib import FrameParams, FrameResult
    lib = C.CDLL(args.lib)
    vp, i32 = C.c_void_p, C.c_int
    u8p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    lib.ofps_hip_init.argtypes = [i32, C.POINTER(vp)]
    lib.ofps_hip_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.ofps_hip_reset_frames.argtypes = [vp]
    lib.ofps_hip_push_frame_async.argtypes = [vp, u8p, i32, i32, i32, C.POINTER(FrameParams), f32p, f32p, C.POINTER(i32)]
    lib.ofps_hip_frame_wait.argtypes = [vp, i32, C.POINTER(FrameResult)]
    lib.ofps_hip_last_error.restype = C.c_char_p
    lib.ofps_hip_last_error.argtypes = [vp]
    h = vp(0)
    assert lib.ofps_hip_init(0, C.byref(h)) == 0
    W, H, B, R = 1920, 1080, 16, 16
    nblk = (W // B) * (H // B)

    def pinned(nbytes):
        p = vp(0)
        assert lib.ofps_hip_host_alloc(h, nbytes, C.byref(p)) == 0
        return p.value
    pin_p = [pinned(W * H) for _ in range(3)]
    ent_p = [pinned(nblk * 16) for _ in range(2)]
    pins = [np.ctypeslib.as_array(C.cast(p, u8p), shape=(H, W)) for p in pin_p]
    rng = np.random.default_rng(1)
    src = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(4)]
    for k in range(3):
        np.copyto(pins[k], src[k])
    prm = FrameParams(B, R, 0, 0.05, 3, 0.003, 0, 16 / 9, 22.275, 0, 200, 0.05, 1000, 0)
    res = FrameResult()
    pc = time.perf_counter

    def check(rc):
        if rc != 0:
            raise RuntimeError(lib.ofps_hip_last_error(h).decode())

    def push(k, e):
        t = i32(0)
        check(lib.ofps_hip_push_frame_async(h, C.cast(pin_p[k % 3], u8p), W, H, W, C.byref(prm), C.cast(ent_p[e], f32p), None, C.byref(t)))
        return t.value

    def loop(n, mode, acc):
        check(lib.ofps_hip_reset_frames(h))
        prev = None
        for k in range(n):
            t0 = pc()
            t = push(k, 0 if mode == "sync" else k % 2)
            t1 = pc()
            if mode == "fill":
                np.copyto(pins[(k + 1) % 3], src[(k + 1) % 4])
            t2 = pc()
            if mode == "sync":
                check(lib.ofps_hip_frame_wait(h, t, C.byref(res)))
            elif prev is not None:
                check(lib.ofps_hip_frame_wait(h, prev, C.byref(res)))
            t3 = pc()
            prev = t
            acc["push"] += t1 - t0; acc["fill"] += t2 - t1; acc["wait"] += t3 - t2
            acc["max_wait"] = max(acc["max_wait"], t3 - t2); acc["max_push"] = max(acc["max_push"], t1 - t0)
        if mode != "sync":
            check(lib.ofps_hip_frame_wait(h, prev, C.byref(res)))

    N = args.frames
    th0 = throttle()
    out = {"lib": os.path.relpath(args.lib, ROOT), "torch_imported": not args.no_torch, "frames": N, "reps": args.reps}
    rows = {m: [] for m in ("sync", "ahead", "fill")}
    for m in rows:
        loop(20, m, {"push": 0, "fill": 0, "wait": 0, "max_wait": 0, "max_push": 0})
    for _ in range(args.reps):
        for m in rows:
            acc = {"push": 0.0, "fill": 0.0, "wait": 0.0, "max_wait": 0.0, "max_push": 0.0}
            t0 = pc()
            loop(N, m, acc)
            el = pc() - t0
            rows[m].append({"ms_per_frame": round(el / N * 1e3, 4), "push": round(acc["push"] / N * 1e3, 4), "wait": round(acc["wait"] / N * 1e3, 4),
                            "fill": round(acc["fill"] / N * 1e3, 4), "max_wait_ms": round(acc["max_wait"] * 1e3, 3),
                            "max_push_ms": round(acc["max_push"] * 1e3, 3)})
    th1 = throttle()
    for m, r in rows.items():
        v = sorted(x["ms_per_frame"] for x in r)
        out[m] = {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "runs": r}
    out["cgroup_throttle_delta"] = {k: th1[k] - th0[k] for k in th0} if th0 else None
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "ofps_amd", "libofps_hip.so"))
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--wrapped", action="store_true", help="bench.py's loop: ofps_amd.runtime.HipContext wrappers instead of raw ctypes")
    ap.add_argument("--content", default="random", choices=["random", "synth"])
    ap.add_argument("--delay-wait-us", type=float, default=0.0, help="busy-wait between the push and the wait")
    ap.add_argument("--delay-push-us", type=float, default=0.0, help="busy-wait in front of the push")
    ap.add_argument("--sweep", action="store_true", help="product library: wrapped / raw x content x delays")
    args = ap.parse_args()
    if not args.all:
        return one(args)
    libs = [("product", os.path.join(ROOT, "ofps_amd", "libofps_hip.so"))]
    ab = os.path.join(ROOT, "build", "ab")
    if os.path.isdir(ab):
        libs += [(n, os.path.join(ab, n, "libofps_hip.so")) for n in sorted(os.listdir(ab)) if os.path.exists(os.path.join(ab, n, "libofps_hip.so"))]
    for name, lib in libs:
        for no_torch in (False, True):
            cmd = [sys.executable, os.path.abspath(__file__), "--lib", lib, "--frames", str(args.frames), "--reps", str(args.reps)] + \
                  (["--no-torch"] if no_torch else [])
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(json.dumps({"variant": name, "no_torch": no_torch, "error": (p.stderr or p.stdout)[-400:]}), flush=True)
                continue
            d = json.loads(line[-1])
            print(json.dumps({"variant": name, "torch": d["torch_imported"],
                              **{m: {k: d[m][k] for k in ("median", "min", "max")} for m in ("sync", "ahead", "fill")},
                              "ahead_split_of_median_run": sorted(d["ahead"]["runs"], key=lambda r: r["ms_per_frame"])[len(d["ahead"]["runs"]) // 2],
                              "throttle": d["cgroup_throttle_delta"]}), flush=True)


if __name__ == "__main__":
    main()
