#!/usr/bin/env python3
"""Fuzz of the multi-device stream dispatcher (ofps_hip_multi_push_frames_async / _frames_wait) with 1-3 workers on device 0:
batches of random sizes (1-6 frames), random numbers of batches in flight (up to two per worker), stream restarts, pageable and
page-locked sources -- every frame's vectors (bits), island and quaternion (bits) against the synchronous per-frame call on one
plain context replaying the same stream.  usage: multi_fuzz.py [batches] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext, MultiDevice

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)
W, H, B, R = 320, 176, 16, 8
fr = synth.luma_sequence(24, W, H, max_step=4, seed=8)
nblk = (W // B) * (H // B)
ref = HipContext(0)
bad = 0; frames_done = 0
t0 = time.perf_counter()
for nw in (1, 2, 3):
    m = MultiDevice([0] * nw)
    pin_src = HipContext(0)                       # a context to allocate page-locked sources from
    pos = 0                                       # position in the stream since the last restart
    ref.reset_frames(); expect = []               # per frame: the plain context's result
    inflight = []                                 # (ticket, frames array, out array, first stream position)
    def collect():
        global bad, frames_done
        t, fa, out, p0 = inflight.pop(0)
        res = m.frames_wait(t)
        for i, r in enumerate(res):
            e = expect[p0 + i]
            ok = r["have_vectors"] == e["have_vectors"]
            if e["have_vectors"]:
                ok = ok and np.array_equal(out[i].view(np.uint32), e["entries"].view(np.uint32)) and np.array_equal(r["quat"].view(np.uint32), e["quat"].view(np.uint32)) \
                     and (r["motion"] is None) == (e["motion"] is None) and (r["motion"] is None or r["motion"][0] == e["motion"][0])
            bad += not ok; frames_done += 1
    for b in range(NB // 3):
        if rng.random() < 0.05:                   # restart the stream (tickets collected first)
            while inflight: collect()
            m.reset_frames(); ref.reset_frames(); pos = 0; expect = []
        n = int(rng.integers(1, 7))
        idx = [(pos + i) % 24 for i in range(n)]
        if rng.random() < 0.5:
            fa = np.ascontiguousarray(fr[idx])                            # pageable source
        else:
            fa = pin_src.pinned_array((n, H, W), np.uint8); np.copyto(fa, fr[idx])
        for i in idx:
            expect.append(ref.push_frame(fr[i], block=B, search_range=R, want_entries=True))
        out = np.zeros((n, nblk, 4), np.float32)
        inflight.append((m.push_frames_async(fa, block=B, search_range=R, out_entries=out), fa, out, pos))
        pos += n
        while len(inflight) > int(rng.integers(0, 2 * nw)): collect()
    while inflight: collect()
    m.close()
print(f"multi fuzz: {frames_done} frames in batches of 1-6 over 1, 2 and 3 workers (seed {SEED}) in {time.perf_counter() - t0:.1f} s, mismatching frames {bad}")
sys.exit(1 if bad else 0)
