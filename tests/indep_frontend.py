"""An independent NumPy restatement of cv-decoder's frame front-end (resize INTER_LINEAR for 8-bit images, BGR->gray), written
from OpenCV's published definitions without looking at oracle/frontend_oracle.c's loops: whole-array arithmetic, tables first.
Test infrastructure: the second opinion the C oracle is compared with (tests/test_frontend_oracle.py)."""
import numpy as np


def _axis_table(src: int, dst: int, horizontal: bool):
    d = np.arange(dst, dtype=np.float64)
    scale = np.float64(1.0) / (np.float64(dst) / np.float64(src))
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= src - 1
        f[hi] = 0; s[hi] = src - 1
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)      # rint = round half to even = cvRound
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, c0, c1


def resize_linear(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    a = np.asarray(img, np.uint8)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    H, W, cn = a.shape
    if (dw, dh) == (W, H):
        out = a.copy()
    elif W == 2 * dw and H == 2 * dh:
        q = a.astype(np.int64)
        out = ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        sx, a0, a1 = _axis_table(W, dw, True)
        sy, b0, b1 = _axis_table(H, dh, False)
        q = a.astype(np.int64)
        x1 = np.minimum(sx + 1, W - 1)
        last = (sx + 1 >= W)
        # horizontal pass for every source row (int32 in OpenCV; the values fit)
        hp = q[:, sx, :] * a0[None, :, None] + q[:, x1, :] * a1[None, :, None]
        hp[:, last, :] = q[:, sx[last], :] * 2048
        r0 = np.clip(sy, 0, H - 1); r1 = np.clip(sy + 1, 0, H - 1)
        D0 = hp[r0]; D1 = hp[r1]
        out = ((((b0[:, None, None] * (D0 >> 4)) >> 16) + ((b1[:, None, None] * (D1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def to_gray(img: np.ndarray, order: str = "bgr") -> np.ndarray:
    a = np.asarray(img, np.int64)
    ch = {c: a[:, :, i] for i, c in enumerate(order)}
    return ((ch["b"] * 1868 + ch["g"] * 9617 + ch["r"] * 4899 + 8192) >> 14).astype(np.uint8)


def cv_grid(W: int, H: int, max_w: int = 150, max_h: int = 150):
    w, h = min(max_w, W), min(max_h, H)
    wb = (w, w * H // W)
    hb = (h * W // H, h)
    return wb if wb[0] < hb[0] else hb
