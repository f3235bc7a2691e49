"""`.mvec` files: the reference's on-disk / wire format for motion vectors.

Per frame: u32 LE vector count, then count x [x, y, mx, my] f32 LE; frames back to back, no header
(writer motion-extract/src/main.rs:23-35, reader motion-loader/src/lib.rs:46-66).  Host-side I/O only.
"""
from __future__ import annotations

import struct
from typing import BinaryIO, Iterator

import numpy as np


def write_frame(f: BinaryIO, entries: np.ndarray) -> None:
    e = np.ascontiguousarray(entries, dtype="<f4").reshape(-1, 4)
    f.write(struct.pack("<I", e.shape[0]))
    f.write(e.tobytes())


def read_frames(f: BinaryIO) -> Iterator[np.ndarray]:
    """Yields one [n,4] float32 array per frame until EOF (a truncated frame raises, like read_exact)."""
    while True:
        hdr = f.read(4)
        if len(hdr) == 0:
            return
        if len(hdr) != 4:
            raise EOFError("truncated .mvec frame header")
        (n,) = struct.unpack("<I", hdr)
        buf = f.read(16 * n)
        if len(buf) != 16 * n:
            raise EOFError("truncated .mvec frame")
        yield np.frombuffer(buf, dtype="<f4").reshape(n, 4).astype(np.float32)
