#!/usr/bin/env python3
"""Third stage of the read-ahead hunt: bench.py's own end_to_end_leg, called three times in ONE process
(is only the first call slow?), and the time course of a fresh context's first frames in 20-frame chunks (mode `course`)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, B, R = 1920, 1080, 16, 16


def main():
    from ofps_amd import synth
    fr = synth.luma_sequence(5, W, H, max_step=R)
    if len(sys.argv) > 1 and sys.argv[1] == "course":
        from ofps_amd.runtime import HipContext
        t_start = time.perf_counter()
        ctx = HipContext(0)
        nblk = (W // B) * (H // B)
        pins = [ctx.pinned_frame(H, W) for _ in range(3)]
        ents = [ctx.pinned_array((nblk, 4)) for _ in range(2)]
        for k in range(3):
            np.copyto(pins[k], fr[k][:, :W])
        kw = dict(block=B, search_range=R, detector=False, estimator=False)
        mode = sys.argv[2] if len(sys.argv) > 2 else "ahead"
        pre_sync = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        ctx.reset_frames()
        for k in range(pre_sync):
            ctx.frame_wait(ctx.push_frame_async(pins[k % 3], out_entries=ents[0], **kw))
        ctx.reset_frames()
        prev = None
        chunks = []
        import gc
        gcmode = os.environ.get("GCMODE", "default")          # default | freeze (gc.collect + gc.freeze first) | disable
        if gcmode == "freeze":
            gc.collect(); gc.freeze()
        elif gcmode == "disable":
            gc.disable()
        events = []                                          # [generation, frame index, ms]

        def on_gc(phase, info):
            if phase == "start":
                events.append([info["generation"], kk[0], time.perf_counter()])
            else:
                events[-1][2] = round((time.perf_counter() - events[-1][2]) * 1e3, 3)
        kk = [0]
        gc.callbacks.append(on_gc)
        t0 = time.perf_counter()
        for k in range(2000):
            kk[0] = k
            t = ctx.push_frame_async(pins[k % 3], out_entries=ents[k % 2], **kw)
            if mode == "sync":
                ctx.frame_wait(t)
            else:
                if prev is not None:
                    ctx.frame_wait(prev)
                prev = t
            if k % 20 == 19:
                t1 = time.perf_counter()
                chunks.append(round((t1 - t0) / 20 * 1e3, 4))
                t0 = t1
        if prev is not None:
            ctx.frame_wait(prev)
        gc.callbacks.remove(on_gc)
        slow = {i: c for i, c in enumerate(chunks) if c > 1.5 * sorted(chunks)[len(chunks) // 2]}
        print(json.dumps({"mode": mode, "pre_sync_frames": pre_sync, "gc": gcmode, "median_chunk": sorted(chunks)[len(chunks) // 2],
                          "slow_chunks_index_to_ms_per_frame": slow,
                          "gc_events_generation_frame_ms": [e for e in events if e[0] >= 1 or e[2] > 0.5],
                          "gen0_collections": sum(1 for e in events if e[0] == 0)}), flush=True)
        ctx.close()
        return
    import bench
    for i in range(3):
        e = bench.end_to_end_leg(fr, W, H, B, R, 0)
        print(json.dumps({"call": i, **{k: e[k]["ms_per_frame"] for k in ("sync", "read_ahead", "read_ahead_with_host_copy")}}), flush=True)


if __name__ == "__main__":
    main()
