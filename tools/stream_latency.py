#!/usr/bin/env python3
"""BASELINE configs[4]: 1080p@60 live stream over TCP, SAD decoder + block-motion detector + Almeida estimator
fused per frame (ofps_hip_push_frame); steady-state latency = frame fully received -> island + quaternion on the
host.  The feeder thread paces raw luma frames (8-byte header: u32 W, u32 H, then W*H bytes) through a loop-back
socket, the way ofps::utils::open_file("tcp://...") feeds a decoder (ofps/src/utils.rs:92-118).  Prints JSON."""
import argparse
import json
import os
import socket
import struct
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofps_amd import synth  # noqa: E402
from ofps_amd.runtime import HipContext  # noqa: E402


def recv_exact(sock, n, buf):
    view = memoryview(buf)[:n]
    got = 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise EOFError
        got += r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--fps", type=float, default=60.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--ransac", action="store_true")
    args = ap.parse_args()
    W, H = args.width, args.height
    clip = synth.luma_sequence(16, W, H, max_step=16)            # looped
    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    port = srv.getsockname()[1]

    def feeder():
        c, _ = srv.accept()
        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        t_next = time.perf_counter()
        for k in range(args.frames):
            now = time.perf_counter()
            if now < t_next:
                time.sleep(t_next - now)
            t_next += 1.0 / args.fps
            c.sendall(struct.pack("<II", W, H)); c.sendall(clip[k % len(clip)].tobytes())
        c.close()

    th = threading.Thread(target=feeder, daemon=True); th.start()
    sock = socket.create_connection(("127.0.0.1", port))
    ctx = HipContext(0)
    # the receive buffer IS the page-locked staging buffer (recv_into writes the frame where the DMA engine reads it):
    # no host copy, no pageable-memory staging inside hipMemcpy
    pinned = ctx.pinned_frame(H, W)
    buf = memoryview(pinned).cast("B")
    lat, proc = [], []
    islands = 0
    for k in range(args.frames):
        hdr = bytearray(8); recv_exact(sock, 8, hdr)
        w, h = struct.unpack("<II", hdr)
        recv_exact(sock, w * h, buf)
        t_arr = time.perf_counter()
        frame = pinned
        r = ctx.push_frame(frame, block=16, search_range=16, use_ransac=args.ransac, seed=k)
        t_done = time.perf_counter()
        if k >= 10:                                               # steady state
            lat.append((t_done - t_arr) * 1e3)
        islands += r["motion"] is not None
    lat = np.array(lat)
    print(json.dumps({"config": f"{W}x{H}@{args.fps:g} TCP loop-back, 16x16 +-16 SAD + block-motion + almeida "
                                f"({'RANSAC' if args.ransac else 'LSQ'}) per frame", "frames": args.frames,
                      "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "p90": round(float(np.percentile(lat, 90)), 3),
                                     "p99": round(float(np.percentile(lat, 99)), 3), "max": round(float(lat.max()), 3)},
                      "frame_budget_ms": round(1e3 / args.fps, 2), "frames_with_motion_island": int(islands)}))


if __name__ == "__main__":
    main()
