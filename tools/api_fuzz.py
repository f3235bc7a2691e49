#!/usr/bin/env python3
"""Stateful fuzz of the C ABI on one context: a random sequence of entry points with random (valid) geometries and parameters --
pair searches, dense flows, the hip_lk decoder (pair, synchronous stream, read-ahead stream, with geometry changes in mid-stream),
the fused per-frame stream (synchronous and read-ahead), densify / interpolate / detect, the Almeida solver (LSQ, RANSAC), the
contrast mask -- every result compared bit for bit with the same call on a context that has done nothing else (fresh context per
stateless call; a dedicated replay context per stream).  Looks for state leaking from one call into another: shared scratch slots,
flags and tags that outlive a call, tickets, rings.  usage: api_fuzz.py [ops] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth
from ofps_amd.runtime import HipContext

OPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)
ctx = HipContext(0)                       # the context under test: everything goes through it
clean = HipContext(0)                     # stateless calls: the same call here (it only ever runs stateless calls, one kind after another --
#                                           itself a weaker version of the test, so a third context replays a sample)
GEOMS = [(64, 48), (160, 96), (320, 176), (97, 61), (384, 216)]
frames = {g: synth.luma_sequence(6, g[0], g[1], max_step=3, seed=40 + g[0]) for g in GEOMS}
bgr_frames = {g: np.clip(frames[g][..., None].astype(int) + np.array([9, -6, 15]) + np.random.default_rng(7).integers(-3, 4, frames[g].shape + (3,)), 0, 255).astype(np.uint8) for g in GEOMS}
def front_end():
    """round 6: a random frame format / "Process Fullres" setting for the dense decoders -> (kwargs, which frame set)"""
    fmt = int(rng.integers(2)); red = bool(rng.integers(2))
    return dict(fmt=fmt, reduced=red), (bgr_frames if fmt else frames)
fields = {n: synth.rotation_field(*n) for n in [(16, 9), (40, 30), (64, 36)]}
bad = []
def same(a, b): return a.shape == b.shape and np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)

# stream state mirrored on replay contexts
lk_stream = {"ctx": HipContext(0), "geom": None, "k": 0, "pending": []}      # read-ahead LK stream on ctx; replay = synchronous pairs
pf_stream = {"ctx": HipContext(0), "geom": None, "k": 0}

def op_sad():
    g = GEOMS[rng.integers(len(GEOMS))]; B, R = [(16, 16), (16, 8), (8, 8), (8, 16)][rng.integers(4)]
    a, b = rng.integers(6, size=2)
    r = ctx.sad_flow(frames[g][a], frames[g][b], B, R); e = clean.sad_flow(frames[g][a], frames[g][b], B, R)
    return same(r, e)
def op_lk_flow():
    g = GEOMS[rng.integers(len(GEOMS))]; L, Rr, it = int(rng.integers(1, 4)), int([2, 3, 4, 6][rng.integers(4)]), int(rng.integers(1, 4))
    a, b = rng.integers(6, size=2)
    return same(ctx.lk_flow(frames[g][a], frames[g][b], L, Rr, it), clean.lk_flow(frames[g][a], frames[g][b], L, Rr, it))
def op_lk_decode():
    g = GEOMS[rng.integers(len(GEOMS))]; a, b = rng.integers(6, size=2)
    fe, fs = front_end()
    kw = dict(contrast_mask=bool(rng.integers(2)), fullres_records=bool(rng.integers(2)) and not fe["reduced"], max_w=int([40, 150][rng.integers(2)]), farneback=bool(rng.integers(2)), **fe)
    r, e = ctx.lk_decode(fs[g][a], fs[g][b], 2, 4, 2, **kw), clean.lk_decode(fs[g][a], fs[g][b], 2, 4, 2, **kw)
    return r[1] == e[1] and same(r[0], e[0])
def op_lk_stream():
    s = lk_stream
    if s["geom"] is None or rng.random() < 0.08:                         # (re)start, maybe with a new geometry, tickets in flight collected first
        for t, *_ in s["pending"]: ctx.lk_frame_wait(t)
        s["pending"] = []; ctx.lk_reset(); s["geom"] = GEOMS[rng.integers(len(GEOMS))]; s["k"] = 0
        s["fe"], s["frames"] = front_end()
        gw_, gh_ = s["geom"]
        s["pins"] = [ctx.pinned_frame(gh_, gw_ * 3).reshape(gh_, gw_, 3) if s["fe"]["fmt"] else ctx.pinned_frame(gh_, gw_) for _ in range(3)]
        # half of the streams are hip_flow streams (round 5): their frames reuse the previous frame's pyramid + expansion, which every
        # other Farneback call through this context (op_fb_flow, op_lk_decode) must invalidate
        s["kw"] = dict(farneback=True, **s["fe"]) if rng.random() < 0.5 else dict(s["fe"])
        s["par"] = (int(rng.integers(1, 4)), int([2, 4, 6][rng.integers(3)]), 2) if s["kw"].get("farneback") else (2, 4, 2)
        # half of those chain their flows (OFPS_HIP_FLOW_USE_PREVIOUS = cv-decoder's OPTFLOW_USE_INITIAL_FLOW): the expected records then come
        # from the same stream run synchronously on the replay context, which does nothing else
        s["chained"] = bool(s["kw"].get("farneback")) and rng.random() < 0.5
        if s["chained"]: s["kw"]["use_previous"] = True; s["ctx"].lk_reset()
    g, k = s["geom"], s["k"]
    fset = s["frames"]
    np.copyto(s["pins"][k % 3], fset[g][k % 6])
    t = ctx.lk_push_frame_async(s["pins"][k % 3], *s["par"], **s["kw"])
    exp = s["ctx"].lk_push_frame(fset[g][k % 6], *s["par"], **s["kw"]) if s["chained"] else None
    s["pending"].append((t, k, exp)); s["k"] += 1
    ok = True
    while len(s["pending"]) > int(rng.integers(1, 3)) - 1 and s["pending"]:      # keep 0 or 1 tickets in flight
        t0, k0, e0 = s["pending"].pop(0)
        r = ctx.lk_frame_wait(t0)
        if k0 == 0: ok &= r is None
        else:
            e = e0 if s["chained"] else clean.lk_decode(s["frames"][g][(k0 - 1) % 6], s["frames"][g][k0 % 6], *s["par"], **s["kw"])
            ok &= r is not None and e is not None and same(r[0], e[0])
    return ok
def op_fb_flow():
    g = GEOMS[rng.integers(len(GEOMS))]; a, b = rng.integers(6, size=2)
    kw = dict(levels=int(rng.integers(0, 5)), winsize=int([5, 9, 13, 15][rng.integers(4)]), iters=int(rng.integers(1, 4)), poly_n=int([5, 7][rng.integers(2)]),
              poly_sigma=float([1.1, 1.5][rng.integers(2)]))
    return same(ctx.farneback_flow(frames[g][a], frames[g][b], **kw), clean.farneback_flow(frames[g][a], frames[g][b], **kw))
def op_push_frame():
    s = pf_stream
    if s["geom"] is None or rng.random() < 0.08:
        ctx.reset_frames(); s["ctx"].reset_frames(); s["geom"] = GEOMS[rng.integers(3)]; s["k"] = 0
    g, k = s["geom"], s["k"]; s["k"] += 1
    kw = dict(block=16, search_range=8, use_ransac=bool(k % 3 == 2), num_iters=30, num_samples=200, seed=k, want_entries=True)
    r = ctx.push_frame(frames[g][k % 6], **kw); e = s["ctx"].push_frame(frames[g][k % 6], **kw)
    if not r["have_vectors"]: return not e["have_vectors"]
    return same(r["entries"], e["entries"]) and same(r["quat"], e["quat"]) and (r["motion"] is None) == (e["motion"] is None) and \
           (r["motion"] is None or r["motion"][0] == e["motion"][0])
def op_densify():
    n = int(rng.integers(1, 900)); w, h = [(16, 9), (40, 30), (150, 84)][rng.integers(3)]
    e = np.zeros((n, 4), np.float32); e[:, :2] = rng.uniform(0, 1, (n, 2)); e[:, 2:] = rng.uniform(-0.02, 0.02, (n, 2))
    ok = same(ctx.densify(e, w, h), clean.densify(e, w, h))
    if w * h <= 1200: ok &= same(ctx.densify_interpolated(e, w, h), clean.densify_interpolated(e, w, h))
    return ok
def op_detect():
    f = fields[list(fields)[rng.integers(3)]].copy(); f[:, 2:] *= rng.uniform(0.5, 3.0)
    r, e = ctx.detect(f), clean.detect(f)
    return (r is None) == (e is None) and (r is None or (r[0] == e[0] and same(r[1], e[1])))
def op_almeida():
    f = fields[list(fields)[rng.integers(3)]]; ran = bool(rng.integers(2)); seed = int(rng.integers(1000))
    r = ctx.almeida(f, 16 / 9, 22.275, use_ransac=ran, num_iters=25, num_samples=150, seed=seed)[0]
    e = clean.almeida(f, 16 / 9, 22.275, use_ransac=ran, num_iters=25, num_samples=150, seed=seed)[0]
    return same(np.asarray(r, np.float32), np.asarray(e, np.float32))
def op_mask():
    g = GEOMS[rng.integers(len(GEOMS))]; a = rng.integers(6)
    return same(ctx.contrast_mask(frames[g][a]), clean.contrast_mask(frames[g][a]))

ops = [op_sad, op_lk_flow, op_lk_decode, op_fb_flow, op_lk_stream, op_lk_stream, op_push_frame, op_push_frame, op_densify, op_detect, op_almeida, op_mask]
t0 = time.perf_counter()
hist = {}
for i in range(OPS):
    f = ops[rng.integers(len(ops))]
    hist[f.__name__] = hist.get(f.__name__, 0) + 1
    if not f(): bad.append((i, f.__name__))
print(f"api fuzz: {OPS} calls (seed {SEED}) in {time.perf_counter() - t0:.1f} s, {hist}, mismatches {len(bad)} {bad[:8]}, "
      f"lk tiles computed by a waiting child {ctx.lk_helped_tiles()}, hip_flow frames that reused the previous expansion {ctx.flow_cache_hits()}")
sys.exit(1 if bad else 0)
