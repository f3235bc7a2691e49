#!/usr/bin/env python3
"""Second stage of the read-ahead hunt.  tools/read_ahead_probe.py showed every library variant -- the tree's,
round 3's, three host-allocation flag sets, three polling budgets -- at 0.055 ms per frame through a raw-ctypes loop while
bench.py's end_to_end leg reports 0.175 on the same box with the same library: the difference is in the loop, not the library.
This script runs bench.py's loop (ofps_amd.runtime.HipContext wrappers) and the raw loop side by side and varies one thing at a
time: the order of the legs, the frame content, a busy-wait in front of the push / in front of the wait, and which page-locked
buffers the loop cycles through.  One JSON line per experiment; every number is the median of `reps` repeats of 300 frames."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, B, R = 1920, 1080, 16, 16


def busy(us):
    if us > 0:
        t = time.perf_counter() + us * 1e-6
        while time.perf_counter() < t:
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--frames", type=int, default=300)
    args = ap.parse_args()
    from ofps_amd import synth
    from ofps_amd.runtime import HipContext
    nblk = (W // B) * (H // B)
    N = args.frames
    kw = dict(block=B, search_range=R, detector=False, estimator=False)

    def setup(content):
        ctx = HipContext(0)
        if content == "synth":
            fr = synth.luma_sequence(5, W, H, max_step=R)
            src = [np.ascontiguousarray(f[:, :W]).copy() for f in fr[:4]]
        else:
            rng = np.random.default_rng(1)
            src = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(4)]
        pins = [ctx.pinned_frame(H, W) for _ in range(3)]
        ents = [ctx.pinned_array((nblk, 4)) for _ in range(2)]
        for k in range(3):
            np.copyto(pins[k], src[k])
        return ctx, src, pins, ents

    def run_sync(ctx, pins, ents, n):
        ctx.reset_frames()
        for k in range(n):
            ctx.frame_wait(ctx.push_frame_async(pins[k % 3], out_entries=ents[0], **kw))

    def run_ahead(ctx, src, pins, ents, n, fill=False, d_push=0.0, d_wait=0.0, npins=3):
        ctx.reset_frames()
        prev = None
        for k in range(n):
            busy(d_push)
            t = ctx.push_frame_async(pins[k % npins], out_entries=ents[k % 2], **kw)
            if fill:
                np.copyto(pins[(k + 1) % 3], src[(k + 1) % 4])
            busy(d_wait)
            if prev is not None:
                ctx.frame_wait(prev)
            prev = t
        ctx.frame_wait(prev)

    def med(fn):
        fn(20)
        v = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn(N)
            v.append((time.perf_counter() - t0) / N * 1e3)
        v.sort()
        return {"median": round(v[len(v) // 2], 4), "min": round(v[0], 4), "max": round(v[-1], 4)}

    def emit(name, **kv):
        print(json.dumps({"experiment": name, **kv}), flush=True)

    # A. bench.py's leg as it is written: sync, then read-ahead, then read-ahead + fill, synth frames
    for content in ("synth", "random"):
        ctx, src, pins, ents = setup(content)
        emit("bench_order", content=content,
             sync=med(lambda n: run_sync(ctx, pins, ents, n)),
             ahead=med(lambda n: run_ahead(ctx, src, pins, ents, n)),
             fill=med(lambda n: run_ahead(ctx, src, pins, ents, n, fill=True)),
             ahead_again=med(lambda n: run_ahead(ctx, src, pins, ents, n)))
        ctx.close()
    # B. read-ahead first, in a fresh context
    for content in ("synth", "random"):
        ctx, src, pins, ents = setup(content)
        emit("ahead_first", content=content, ahead=med(lambda n: run_ahead(ctx, src, pins, ents, n)),
             sync=med(lambda n: run_sync(ctx, pins, ents, n)), ahead_again=med(lambda n: run_ahead(ctx, src, pins, ents, n)))
        ctx.close()
    # C. delays (synth content)
    ctx, src, pins, ents = setup("synth")
    for d in (0, 5, 10, 20, 40, 80):
        emit("delay_before_wait_us", us=d, ahead=med(lambda n: run_ahead(ctx, src, pins, ents, n, d_wait=d)))
    for d in (5, 10, 20, 40, 80):
        emit("delay_before_push_us", us=d, ahead=med(lambda n: run_ahead(ctx, src, pins, ents, n, d_push=d)))
    # D. one source buffer instead of three
    emit("one_pinned_source", ahead=med(lambda n: run_ahead(ctx, src, pins, ents, n, npins=1)))
    # E. where the time goes in the slow loop: per-call times of 300 frames
    ctx.reset_frames()
    pc = time.perf_counter
    tp, tw = [], []
    prev = None
    for k in range(N):
        t0 = pc()
        t = ctx.push_frame_async(pins[k % 3], out_entries=ents[k % 2], **kw)
        t1 = pc()
        if prev is not None:
            ctx.frame_wait(prev)
        t2 = pc()
        prev = t
        tp.append(t1 - t0); tw.append(t2 - t1)
    ctx.frame_wait(prev)
    tp, tw = np.array(tp[10:]) * 1e3, np.array(tw[10:]) * 1e3
    emit("call_times_ms", push={"mean": round(float(tp.mean()), 4), "p50": round(float(np.median(tp)), 4), "p99": round(float(np.percentile(tp, 99)), 4)},
         wait={"mean": round(float(tw.mean()), 4), "p50": round(float(np.median(tw)), 4), "p99": round(float(np.percentile(tw, 99)), 4)},
         first_20_push=[round(float(x), 3) for x in tp[:20]], first_20_wait=[round(float(x), 3) for x in tw[:20]])
    ctx.close()


if __name__ == "__main__":
    main()
