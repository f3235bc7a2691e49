"""An INDEPENDENT float64 restatement of the Farneback flow the oracle restates (oracle/farneback_oracle.c), written with
scipy.ndimage building blocks instead of loops: correlate1d for the separable filters, uniform_filter for the window sums,
fancy indexing for the resizes and the warp.  It shares no code with oracle/ or ofps_amd/; an index slip, a swapped channel or a
wrong border rule in the C restatement cannot cancel out against it.  Differences are float32-vs-float64 rounding only."""
import numpy as np
from scipy import ndimage


def round_half_even(v):
    return int(np.rint(v))


def layers(W, H, levels=5):
    scale, k = 1.0, 0
    while k < levels:
        scale *= 0.5
        if W * scale < 32 or H * scale < 32:
            break
        k += 1
    return k


def blur_taps(k):
    scale = 0.5 ** k
    sigma = (1.0 / scale - 1.0) * 0.5
    ksize = max(round_half_even(sigma * 5) | 1, 3)
    if sigma <= 0:
        return np.array([0.25, 0.5, 0.25])
    x = np.arange(ksize) - (ksize - 1) * 0.5
    t = np.exp(-0.5 / sigma ** 2 * x * x).astype(np.float32).astype(np.float64)
    return (t / t.sum()).astype(np.float32).astype(np.float64)


def resize_axis(dn, sn):
    scale = 1.0 / (dn / sn)
    f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32).astype(np.float64)
    s = np.floor(f).astype(int)
    f = f - s
    f[s < 0] = 0; s[s < 0] = 0
    f[s >= sn - 1] = 0; s[s >= sn - 1] = sn - 1
    return s, np.minimum(s + 1, sn - 1), f


def resize_linear(img, dw, dh):
    sh, sw = img.shape[:2]
    x0, x1, fx = resize_axis(dw, sw)
    y0, y1, fy = resize_axis(dh, sh)
    fx = fx.reshape((1, dw) + (1,) * (img.ndim - 2)); fy = fy.reshape((dh, 1) + (1,) * (img.ndim - 2))
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def layer_image(img, k, w, h):
    t = blur_taps(k)
    f = img.astype(np.float64)
    f = ndimage.correlate1d(f, t, axis=1, mode="mirror")
    f = ndimage.correlate1d(f, t, axis=0, mode="mirror")
    return f if f.shape == (h, w) else resize_linear(f, w, h)


def poly_kernels(n, sigma):
    x = np.arange(-n, n + 1)
    g = np.exp(-x * x / (2 * sigma * sigma)).astype(np.float32).astype(np.float64)
    g = (g / g.sum()).astype(np.float32).astype(np.float64)
    xs, ys = np.meshgrid(x, x)
    gg = np.outer(g, g)
    basis = [np.ones_like(xs), xs, ys, xs * xs, ys * ys, xs * ys]
    G = np.array([[(gg * a * b).sum() for b in basis] for a in basis], np.float64)     # the 6x6 moment matrix, inverted numerically
    iG = np.linalg.inv(G)
    return g, x * g, x * x * g, (iG[1, 1], iG[0, 3], iG[3, 3], iG[5, 5])


def poly_exp(I, n=7, sigma=1.5):
    g, xg, xxg, (ig11, ig03, ig33, ig55) = poly_kernels(n, sigma)
    v0 = ndimage.correlate1d(I, g, axis=0, mode="nearest")          # smoothed in y
    v1 = ndimage.correlate1d(I, xg, axis=0, mode="nearest")         # y derivative moment
    v2 = ndimage.correlate1d(I, xxg, axis=0, mode="nearest")
    b1 = ndimage.correlate1d(v0, g, axis=1, mode="nearest")
    b2 = ndimage.correlate1d(v0, xg, axis=1, mode="nearest")
    b4 = ndimage.correlate1d(v0, xxg, axis=1, mode="nearest")
    b3 = ndimage.correlate1d(v1, g, axis=1, mode="nearest")
    b6 = ndimage.correlate1d(v1, xg, axis=1, mode="nearest")
    b5 = ndimage.correlate1d(v2, g, axis=1, mode="nearest")
    return np.stack([b3 * ig11, b2 * ig11, b1 * ig03 + b5 * ig33, b1 * ig03 + b4 * ig33, b6 * ig55], -1)


def matrices(R0, R1, flow):
    h, w = flow.shape[:2]
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    dx, dy = flow[..., 0], flow[..., 1]
    fx = (xx + dx).astype(np.float32).astype(np.float64); fy = (yy + dy).astype(np.float32).astype(np.float64)
    x1 = np.floor(fx).astype(int); y1 = np.floor(fy).astype(int)
    fx = fx - x1; fy = fy - y1
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc = np.clip(x1, 0, w - 2); yc = np.clip(y1, 0, h - 2)
    a00 = ((1 - fx) * (1 - fy))[..., None]; a01 = (fx * (1 - fy))[..., None]; a10 = ((1 - fx) * fy)[..., None]; a11 = (fx * fy)[..., None]
    S = a00 * R1[yc, xc] + a01 * R1[yc, xc + 1] + a10 * R1[yc + 1, xc] + a11 * R1[yc + 1, xc + 1]
    r2 = np.where(inside, S[..., 0], 0.0); r3 = np.where(inside, S[..., 1], 0.0)
    r4 = np.where(inside, (R0[..., 2] + S[..., 2]) * 0.5, R0[..., 2])
    r5 = np.where(inside, (R0[..., 3] + S[..., 3]) * 0.5, R0[..., 3])
    r6 = np.where(inside, (R0[..., 4] + S[..., 4]) * 0.25, R0[..., 4] * 0.5)
    r2 = (R0[..., 0] - r2) * 0.5; r3 = (R0[..., 1] - r3) * 0.5
    r2 = r2 + r4 * dy + r6 * dx
    r3 = r3 + r6 * dy + r5 * dx
    border = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472], np.float32).astype(np.float64)
    sx = np.ones(w); sy = np.ones(h)
    for i in range(min(5, w)):
        sx[i] *= border[i]; sx[w - 1 - i] *= border[i]
    for i in range(min(5, h)):
        sy[i] *= border[i]; sy[h - 1 - i] *= border[i]
    sc = sy[:, None] * sx[None, :]
    r2, r3, r4, r5, r6 = r2 * sc, r3 * sc, r4 * sc, r5 * sc, r6 * sc
    return np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], -1)


def update_flow(M, winsize):
    B = np.stack([ndimage.uniform_filter(M[..., c], winsize, mode="nearest") for c in range(5)], -1)     # mean = sum / winsize^2
    g11, g12, g22, h1, h2 = [B[..., c] for c in range(5)]
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    return np.stack([(g11 * h2 - g12 * h1) * idet, (g22 * h1 - g12 * h2) * idet], -1)


def farneback(prev, cur, levels=5, winsize=13, iters=3, poly_n=7, poly_sigma=1.5):
    H, W = prev.shape
    L = layers(W, H, levels)
    flow = None
    for k in range(L, -1, -1):
        scale = 0.5 ** k
        w, h = round_half_even(W * scale), round_half_even(H * scale)
        flow = np.zeros((h, w, 2)) if flow is None else resize_linear(flow, w, h) * 2.0
        R0 = poly_exp(layer_image(prev, k, w, h), poly_n, poly_sigma)
        R1 = poly_exp(layer_image(cur, k, w, h), poly_n, poly_sigma)
        M = matrices(R0, R1, flow)
        for i in range(iters):
            flow = update_flow(M, winsize)
            if i < iters - 1:
                M = matrices(R0, R1, flow)
    return flow
