#!/usr/bin/env python3
"""Which 32x8 tiles of the LK level kernel have a current-frame sample rectangle that does not fit LDS, per pyramid level and
Gauss-Newton step, and how many sub-groups a greedy anchor grouping needs for them (CPU only: the oracle's per-step flow
trace, oracle.lk_flow_trace).  usage: lk_tile_stats.py [max_step ...]   (default 3 16)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ofps_amd import synth

R, TX, TY, LW, LH = 4, 32, 8, 63, 49


def boxes(flow, w, h):
    x = np.arange(w)[None, :]; y = np.arange(h)[:, None]
    def org(q, fl, lim):
        return np.clip(np.floor(q.astype(np.float32) + fl), -1, lim).astype(np.int64)
    x0 = org(np.clip(x - R, 0, w - 1), flow[..., 0], w); x1 = org(np.clip(x + R, 0, w - 1), flow[..., 0], w) + 1
    y0 = org(np.clip(y - R, 0, h - 1), flow[..., 1], h); y1 = org(np.clip(y + R, 0, h - 1), flow[..., 1], h) + 1
    return x0, x1, y0, y1


def analyse(flow, cap_rounds=64):
    h, w = flow.shape[:2]
    x0, x1, y0, y1 = boxes(flow, w, h)
    nt = 0; fall = 0; rounds_hist = {}; left_px = {4: 0, 8: 0}; fall_px = 0
    for ty in range(0, h, TY):
        for tx in range(0, w, TX):
            s = np.s_[ty:ty + TY, tx:tx + TX]
            a0, a1, b0, b1 = x0[s].ravel(), x1[s].ravel(), y0[s].ravel(), y1[s].ravel()
            nt += 1
            if a1.max() - a0.min() < LW and b1.max() - b0.min() < LH:
                continue
            fall += 1; fall_px += a0.size
            done = np.zeros(a0.size, bool); r = 0
            while not done.all() and r < cap_rounds:
                i = np.flatnonzero(~done)[0]
                # rectangle anchored at pixel i's window, centred in the capacity
                cx0 = a0[i] - (LW - 1 - (a1[i] - a0[i])) // 2; cy0 = b0[i] - (LH - 1 - (b1[i] - b0[i])) // 2
                ok = ~done & (a0 >= cx0) & (a1 <= cx0 + LW - 1) & (b0 >= cy0) & (b1 <= cy0 + LH - 1)
                done |= ok; r += 1
                if r in left_px: left_px[r] += int((~done).sum())
            for k in left_px:
                pass
            rounds_hist[r] = rounds_hist.get(r, 0) + 1
    return {"tiles": nt, "unfit_tiles": fall, "unfit_px": fall_px, "rounds_hist": dict(sorted(rounds_hist.items())),
            "px_left_after_rounds": left_px}


def main():
    steps = [int(a) for a in sys.argv[1:]] or [3, 16]
    out = {}
    for ms in steps:
        fr = synth.luma_sequence(2, 1920, 1080, max_step=ms, seed=11)
        _, tr = oracle.lk_flow_trace(fr[0], fr[1], 3, 4, 3)
        out[f"pm{ms}"] = {f"L{l}_step{it}": analyse(tr[l][it]) for l in (2, 1, 0) for it in range(3)}
        print(json.dumps({f"pm{ms}": out[f"pm{ms}"]}), flush=True)


if __name__ == "__main__":
    main()
