"""Second, independent float32 NumPy restatement of the reference functions (TEST INFRASTRUCTURE).

Purpose: guard against a self-consistent-but-wrong C oracle (SURVEY.md 8c golden-vector plan).
It is written differently on purpose: vectorised, closed-form camera model instead of generic
4x4 products, scipy labelling instead of a LIFO flood fill, np.add.at-free sequential sums
via sorting.  tests/test_oracle.py cross-checks it against oracle/ofps_oracle.c.

Reference lines restated: ofps/src/motion_field.rs:133-190,297-308; ofps/src/camera.rs:26-161;
block-motion-detector/src/lib.rs:49-118; almeida-estimator/src/lib.rs:123-200.
"""
from __future__ import annotations

import numpy as np

F = np.float32
EPSILON = F(1.1920929e-07)


# ---------------------------------------------------------------- densifier
def cell_index(entries, w, h):
    e = np.asarray(entries, F).reshape(-1, 4)
    px, py = e[:, 0], e[:, 1]
    with np.errstate(invalid="ignore"):
        gt0 = (px > 0) & (py > 0)                 # nalgebra::clamp on Point2: all-components order
        lt1 = (px < 1) & (py < 1)
    cx = np.where(gt0, np.where(lt1, px, F(1)), F(0)).astype(F)
    cy = np.where(gt0, np.where(lt1, py, F(1)), F(0)).astype(F)

    # f32::round (half away from zero) for v >= 0, written without v + 0.5 (which can itself round up in f32)
    vx = (cx * F(w - 1)).astype(F); vy = (cy * F(h - 1)).astype(F)
    x = np.where(vx - np.floor(vx) >= F(0.5), np.floor(vx) + 1, np.floor(vx)).astype(np.int64)
    y = np.where(vy - np.floor(vy) >= F(0.5), np.floor(vy) + 1, np.floor(vy)).astype(np.int64)
    return x, y


def densify(entries, w, h):
    """-> field[h,w,2] f32, cells[n,2]; sums in input order per cell (motion_field.rs:141-147)."""
    e = np.asarray(entries, F).reshape(-1, 4)
    x, y = cell_index(e, w, h)
    idx = y * w + x
    order = np.argsort(idx, kind="stable")
    sums = np.zeros((w * h, 2), F)
    cnts = np.full((w * h,), EPSILON, F)
    sidx = idx[order]
    bounds = np.flatnonzero(np.diff(sidx)) + 1
    starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [len(sidx)]])
    for s, t in zip(starts, ends):
        if t <= s:
            continue
        c = sidx[s]
        acc = np.zeros(2, F); cn = EPSILON
        for k in order[s:t]:
            acc = (e[k, 2:4] * F(1.0) + acc).astype(F)
            cn = F(cn + F(1.0))
        sums[c] = acc; cnts[c] = cn
    field = (sums / cnts[:, None]).astype(F)
    return field.reshape(h, w, 2), np.stack([x, y], 1)


# ---------------------------------------------------------------- detector
def block_dim(min_size, subdivide):
    bw = F(np.sqrt(F(min_size))) / F(subdivide)
    return int(np.ceil(F(1.0) / bw))


def detect_motion(entries, min_size=0.05, subdivide=3, target_motion=0.003):
    from scipy import ndimage
    dim = block_dim(min_size, subdivide)
    mf, _ = densify(entries, dim, dim)
    mag = np.sqrt((mf[..., 0] * mf[..., 0] + mf[..., 1] * mf[..., 1]).astype(F)).astype(F)
    mask = mag >= F(target_motion)
    lab, n = ndimage.label(mask, structure=np.ones((3, 3), int))     # 8-connectivity
    if n == 0:
        return None
    areas = ndimage.sum(mask, lab, index=np.arange(1, n + 1)).astype(int)
    # first island in raster order of its first cell wins ties (strict > in lib.rs:106)
    firsts = [np.flatnonzero(lab.ravel() == k)[0] for k in range(1, n + 1)]
    best = max(range(n), key=lambda k: (areas[k], -firsts[k]))
    area = int(areas[best])
    if not (F(area) / F(dim * dim) >= F(min_size)):
        return None
    out = np.zeros_like(mf)
    sel = lab == (best + 1)
    out[sel] = mf[sel]
    sy, sx = divmod(int(firsts[best]), dim)
    out[sy, sx] = 0                      # the seed cell is never copied (lib.rs:79-102)
    return area, out


# ---------------------------------------------------------------- camera (closed form)
class Camera:
    def __init__(self, aspect, fov_y_deg):
        self.aspect = F(aspect); self.fov_y = F(fov_y_deg)
        fovy = F(self.fov_y * F(np.pi / 180.0))
        self.m11 = F(1) / F(np.tan(F(fovy / F(2))))
        self.m00 = F(self.m11 / self.aspect)
        zn, zf = F(0.1), F(10.0)
        self.m22 = F((zf + zn) / (zn - zf))
        self.m23 = F(F(zf * zn) * F(2) / (zn - zf))
        self.r00 = F(1) / self.m00; self.r11 = F(1) / self.m11
        self.r32 = F(1) / self.m23; self.r33 = F(self.m22 * self.r32)

    def delta(self, pos, R):
        """pos[n,2], R 3x3 (f32) -> delta[n,2].  Closed form of camera.rs:89-117: world =
        (-r00*cx, -1, r11*cy)/n0, rotate, view-permute, perspective, divide by NDC z."""
        pos = np.asarray(pos, F).reshape(-1, 2); R = np.asarray(R, F)[:3, :3]
        cx = (pos[:, 0] * F(2) - F(1)).astype(F); cy = (pos[:, 1] * F(2) - F(1)).astype(F)
        n0 = F(self.r32 + self.r33)
        wx = ((-self.r00) * cx / n0).astype(F)
        wy = np.full_like(wx, F(-1) / n0)
        wz = (self.r11 * cy / n0).astype(F)
        rx = ((R[0, 0] * wx + R[0, 1] * wy).astype(F) + R[0, 2] * wz).astype(F)
        ry = ((R[1, 0] * wx + R[1, 1] * wy).astype(F) + R[1, 2] * wz).astype(F)
        rz = ((R[2, 0] * wx + R[2, 1] * wy).astype(F) + R[2, 2] * wz).astype(F)
        px, py, pz = (-rx).astype(F), rz, ry           # view: (-x, z, y)
        inv = (F(-1) / pz).astype(F)
        sx = (self.m00 * px * inv).astype(F); sy = (self.m11 * py * inv).astype(F)
        sz = ((self.m22 * pz + self.m23).astype(F) * inv).astype(F)
        ox = (((sx / sz).astype(F) + F(1)) * F(0.5)).astype(F)
        oy = (((sy / sz).astype(F) + F(1)) * F(0.5)).astype(F)
        return np.stack([ox - pos[:, 0], oy - pos[:, 1]], 1).astype(F)


def rot3_from_euler(roll, pitch, yaw):
    sr, cr = F(np.sin(F(roll))), F(np.cos(F(roll)))
    sp, cp = F(np.sin(F(pitch))), F(np.cos(F(pitch)))
    sy, cy = F(np.sin(F(yaw))), F(np.cos(F(yaw)))
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], F)


def quat_from_euler(roll, pitch, yaw):
    sr, cr = F(np.sin(F(roll) * F(0.5))), F(np.cos(F(roll) * F(0.5)))
    sp, cp = F(np.sin(F(pitch) * F(0.5))), F(np.cos(F(pitch) * F(0.5)))
    sy, cy = F(np.sin(F(yaw) * F(0.5))), F(np.cos(F(yaw) * F(0.5)))
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy], F)


def quat_mul(a, b):
    aw, ai, aj, ak = [F(v) for v in a]; bw, bi, bj, bk = [F(v) for v in b]
    return np.array([aw * bw - ai * bi - aj * bj - ak * bk, aw * bi + ai * bw + aj * bk - ak * bj,
                     aw * bj - ai * bk + aj * bw + ak * bi, aw * bk + ai * bj - aj * bi + ak * bw], F)


def quat_to_rot3(q):
    w, i, j, k = [F(v) for v in q]
    ww, ii, jj, kk = w * w, i * i, j * j, k * k
    ij, wk, wj = i * j * F(2), w * k * F(2), w * j * F(2)
    ik, jk, wi = i * k * F(2), j * k * F(2), w * i * F(2)
    return np.array([[ww + ii - jj - kk, ij - wk, wj + ik],
                     [wk + ij, ww - ii + jj - kk, jk - wi],
                     [ik - wj, wi + jk, ww - ii - jj + kk]], F)


def solve_ypr_given(entries, cam: Camera):
    """almeida-estimator/src/lib.rs:123-200, vectorised (pairwise-summed dots, float64-free)."""
    e = np.asarray(entries, F).reshape(-1, 4)
    EPS = F(F(0.001) * F(np.pi) / F(180.0)); ALPHA = F(0.5)
    limit = int(np.ceil(15.0 / 0.5))
    pos, mot = e[:, 0:2], e[:, 2:4]
    protos = [cam.delta(pos, rot3_from_euler(0, EPS, 0)), cam.delta(pos, rot3_from_euler(EPS, 0, 0)),
              cam.delta(pos, rot3_from_euler(0, 0, -EPS))]
    rot = np.array([1, 0, 0, 0], F)
    for it in range(limit):
        alpha = F(1.0) if it == limit - 1 else ALPHA
        res = (mot - cam.delta(pos, quat_to_rot3(rot))).astype(F)
        A = np.zeros((3, 3), F); b = np.zeros(3, F)
        for r in range(3):
            for c in range(3):
                A[r, c] = np.sum((protos[c] * protos[r]).sum(1, dtype=F), dtype=F)
            b[r] = np.sum((protos[r] * res).sum(1, dtype=F), dtype=F)
        try:
            model = np.linalg.solve(A.astype(np.float64), b.astype(np.float64)).astype(F)
        except np.linalg.LinAlgError:
            model = np.zeros(3, F)
        model = (model * EPS * alpha).astype(F)
        roll = quat_from_euler(0, model[0], 0); pitch = quat_from_euler(model[1], 0, 0)
        yaw = quat_from_euler(0, 0, -model[2])
        rot = quat_mul(rot, quat_mul(quat_mul(pitch, roll), yaw))
    return np.array([rot[0], -rot[1], -rot[2], -rot[3]], F)


# ---------------------------------------------------------------- SAD (N1, build-defined)
def sad_flow(prev, cur, B, R):
    prev = np.asarray(prev, np.int32); cur = np.asarray(cur, np.int32)
    H, W = prev.shape
    nbx, nby = W // B, H // B
    best = np.zeros((nby * nbx, 3), np.int32)
    for by in range(nby):
        for bx in range(nbx):
            x0, y0 = bx * B, by * B
            c = cur[y0:y0 + B, x0:x0 + B]
            bk = None
            for dy in range(-R, R + 1):
                if y0 + dy < 0 or y0 + dy + B > H:
                    continue
                for dx in range(-R, R + 1):
                    if x0 + dx < 0 or x0 + dx + B > W:
                        continue
                    sad = int(np.abs(c - prev[y0 + dy:y0 + dy + B, x0 + dx:x0 + dx + B]).sum())
                    key = (sad, dx * dx + dy * dy, dy + R, dx + R)
                    if bk is None or key < bk:
                        bk = key; best[by * nbx + bx] = (dx, dy, sad)
    return best


def contrast_mask(gray):
    """Second restatement of cv-decoder's contrast mask (cv-decoder/src/lib.rs:203-237) from OpenCV's published
    definitions: Sobel(dx=1, dy=1, ksize 5) = separable [-1,-2,0,2,1] x [-1,-2,0,2,1] with BORDER_REFLECT_101,
    `> 20`, dilation by the 11x11 MORPH_ELLIPSE element (row half-widths cvRound(5*sqrt(1 - dy^2/25)))."""
    g = np.asarray(gray, np.uint8).astype(np.int64)
    H, W = g.shape
    k = np.array([-1, -2, 0, 2, 1], np.int64)
    # reflect101 with repeated reflection for tiny images
    def idx(n, lo, hi):
        i = np.arange(lo, hi)
        if n == 1:
            return np.zeros_like(i)
        period = 2 * n - 2
        i = np.mod(i, period)
        return np.where(i >= n, period - i, i)
    gp = g[idx(H, -2, H + 2)][:, idx(W, -2, W + 2)]
    hx = sum(k[j] * gp[:, j:j + W] for j in range(5))
    s = sum(k[i] * hx[i:i + H, :] for i in range(5))
    thr = (s > 20)
    hw = [int(np.rint(5.0 * np.sqrt((25.0 - dy * dy) / 25.0))) for dy in range(-5, 6)]
    tp = np.zeros((H + 10, W + 10), bool)
    tp[5:5 + H, 5:5 + W] = thr
    out = np.zeros((H, W), bool)
    for i, dy in enumerate(range(-5, 6)):
        for dx in range(-hw[i], hw[i] + 1):
            out |= tp[5 + dy:5 + dy + H, 5 + dx:5 + dx + W]
    return out.astype(np.uint8)


def _lk_pyr_down(img):
    """[1 4 6 4 1]/16 separable blur with replicated border, every second sample (orc lk_pyr_down)."""
    h, w = img.shape
    w1, h1 = (w + 1) // 2, (h + 1) // 2
    xs = np.arange(w1) * 2
    def tap(a, idx, n, axis):
        return np.take(a, np.clip(idx, 0, n - 1), axis=axis)
    t = ((((tap(img, xs - 2, w, 1) + F(4) * tap(img, xs - 1, w, 1)) + F(6) * tap(img, xs, w, 1))
          + F(4) * tap(img, xs + 1, w, 1)) + tap(img, xs + 2, w, 1)) * F(0.0625)
    ys = np.arange(h1) * 2
    o = ((((tap(t, ys - 2, h, 0) + F(4) * tap(t, ys - 1, h, 0)) + F(6) * tap(t, ys, h, 0))
          + F(4) * tap(t, ys + 1, h, 0)) + tap(t, ys + 2, h, 0)) * F(0.0625)
    return o.astype(F)


def fma32(a, b, c):
    """Correctly rounded f32 fused multiply-add on arrays, without a hardware fma: the product of two f32 is exact in f64
    (48 bits), the f64 sum p + c is made ROUND-TO-ODD with the error term of a TwoSum (if the sum was inexact, the result
    whose last mantissa bit is 1 of the two f64 neighbours of the exact value), and a round-to-odd value with 53 >= 2 * 24 + 2
    bits rounds to f32 exactly like the infinitely precise one (no double rounding)."""
    a = np.asarray(a, F).astype(np.float64); b = np.asarray(b, F).astype(np.float64); c = np.asarray(c, F).astype(np.float64)
    a, b, c = np.broadcast_arrays(a, b, c)
    p = a * b                                             # exact
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)                       # exact error of the sum (TwoSum)
    s = np.array(s, np.float64)                           # writable copy
    bits = s.view(np.int64)
    inexact = (err != 0) & np.isfinite(s)
    even = (bits & 1) == 0
    # the exact value lies strictly between s and its neighbour in the direction of err: if s is even, that neighbour is odd
    toward_larger_magnitude = (err > 0) == (s > 0)
    step = np.where(toward_larger_magnitude, 1, -1).astype(np.int64)
    fix = inexact & even & (s != 0)
    bits[fix] += step[fix]
    return s.astype(F)


LK_SPEC_FMA = True          # revision 2 of the build-defined N2 spec (oracle/ofps_oracle.c: ORC_LK_SPEC_FMA)


def _lk_lerp(a, b, t):
    if LK_SPEC_FMA:
        return fma32(t, (b - a).astype(F), a)
    return (a + t * (b - a)).astype(F)


def _lk_bilinear(J, fx, fy):
    h, w = J.shape
    x0f, y0f = np.floor(fx), np.floor(fy)
    ax, ay = (fx - x0f).astype(F), (fy - y0f).astype(F)
    x0 = np.clip(x0f, -1.0, float(w)).astype(np.int64)
    y0 = np.clip(y0f, -1.0, float(h)).astype(np.int64)
    xa, xb = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    ya, yb = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    j00, j10, j01, j11 = J[ya, xa], J[ya, xb], J[yb, xa], J[yb, xb]
    top = _lk_lerp(j00, j10, ax)
    bot = _lk_lerp(j01, j11, ax)
    return _lk_lerp(top, bot, ay)


def lk_flow(prev, cur, levels=3, radius=4, iters=3, init=None):
    """Second restatement of the build-defined pyramidal LK (oracle/ofps_oracle.c:orc_lk_flow): same f32 operations in the
    same order, vectorised over pixels with explicit loops over the window taps -> the same bits."""
    I = [np.asarray(prev, np.uint8).astype(F)]
    J = [np.asarray(cur, np.uint8).astype(F)]
    for _ in range(1, levels):
        I.append(_lk_pyr_down(I[-1])); J.append(_lk_pyr_down(J[-1]))
    flow = None
    for l in range(levels - 1, -1, -1):
        Il, Jl = I[l], J[l]
        h, w = Il.shape
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        if flow is None and init is not None:                # caller-supplied prior for the coarsest level (orc_lk_flow_init)
            u = np.asarray(init, F)[..., 0].copy(); v = np.asarray(init, F)[..., 1].copy()
        elif flow is None:
            u = np.zeros((h, w), F); v = np.zeros((h, w), F)
        else:
            h1, w1 = flow[0].shape
            sy, sx = np.clip(yy // 2, 0, h1 - 1), np.clip(xx // 2, 0, w1 - 1)
            u = (F(2) * flow[0][sy, sx]).astype(F); v = (F(2) * flow[1][sy, sx]).astype(F)
        gx = ((Il[yy, np.clip(xx + 1, 0, w - 1)] - Il[yy, np.clip(xx - 1, 0, w - 1)]) * F(0.5)).astype(F)
        gy = ((Il[np.clip(yy + 1, 0, h - 1), xx] - Il[np.clip(yy - 1, 0, h - 1), xx]) * F(0.5)).astype(F)
        for _ in range(iters):
            gxx = np.zeros((h, w), F); gxy = np.zeros((h, w), F); gyy = np.zeros((h, w), F)
            bx = np.zeros((h, w), F); by = np.zeros((h, w), F)
            for dy in range(-radius, radius + 1):
                for dx in range(-radius, radius + 1):
                    qx, qy = np.clip(xx + dx, 0, w - 1), np.clip(yy + dy, 0, h - 1)
                    ix, iy = gx[qy, qx], gy[qy, qx]
                    d = (Il[qy, qx] - _lk_bilinear(Jl, (qx.astype(F) + u).astype(F), (qy.astype(F) + v).astype(F))).astype(F)
                    if LK_SPEC_FMA:
                        gxx = fma32(ix, ix, gxx); gxy = fma32(ix, iy, gxy); gyy = fma32(iy, iy, gyy)
                        bx = fma32(ix, d, bx); by = fma32(iy, d, by)
                    else:
                        gxx = (gxx + ix * ix).astype(F); gxy = (gxy + ix * iy).astype(F); gyy = (gyy + iy * iy).astype(F)
                        bx = (bx + ix * d).astype(F); by = (by + iy * d).astype(F)
            det = (gxx * gyy - gxy * gxy).astype(F)
            ok = det > F(0.01)
            with np.errstate(divide="ignore", invalid="ignore"):
                du = np.where(ok, ((gyy * bx - gxy * by).astype(F) / det).astype(F), F(0)).astype(F)
                dv = np.where(ok, ((gxx * by - gxy * bx).astype(F) / det).astype(F), F(0)).astype(F)
            u = (u + du).astype(F); v = (v + dv).astype(F)
        flow = (u, v)
    return np.stack(flow, axis=-1).astype(F)


def densify_interpolated(entries, w, h):
    """Second restatement of new_densifier + add_vector (all entries) + interpolate_empty_cells + MotionField::from
    (ofps/src/motion_field.rs:133-147, 164-178, 193-308), written from the reference's text with an ordered set for its
    BTreeSet<InterpCell{neighbors, idx}>; f32 arithmetic through NumPy scalars in the reference's operation order."""
    from sortedcontainers import SortedSet
    e = np.asarray(entries, F).reshape(-1, 4)
    cells = w * h
    summ = np.zeros((cells, 2), F)
    cnt = np.full(cells, np.finfo(F).eps, F)              # Matrix2xX::repeat(.., EPSILON): both rows stay equal
    xs, ys = cell_index(e, w, h)
    one = F(1.0)
    for k in range(e.shape[0]):                           # add_vector: weight 1
        i = int(ys[k]) * w + int(xs[k])
        cnt[i] = F(cnt[i] + one)
        summ[i, 0] = F(F(e[k, 2] * one) + summ[i, 0])
        summ[i, 1] = F(F(e[k, 3] * one) + summ[i, 1])
    nbrs = [(-1, 0), (0, -1), (-1, -1), (1, 0), (0, 1), (1, 1)]

    def calc_counts(i):
        x, y = i % w, i // w
        return sum(1 for ox, oy in nbrs if 0 <= x + ox < w and 0 <= y + oy < h and cnt[(x + ox) + (y + oy) * w] > F(0.1))

    queue = SortedSet((-calc_counts(i), i) for i in range(cells) if cnt[i] < F(0.5))
    if len(queue) != cells:                               # "no motion vectors at all": nothing to interpolate from
        while queue:
            cell = queue.pop(0)
            i = cell[1]
            x, y = i % w, i // w
            added = False
            for ox, oy in nbrs:
                nx, ny = x + ox, y + oy
                if 0 <= nx < w and 0 <= ny < h:
                    idx = nx + ny * w
                    c = cnt[idx]
                    if c > F(0.1):
                        scale = F(one - F(np.sqrt(F(ox * ox + oy * oy)) * F(0.5)))
                        inv_cnt = F(one / c)
                        s = F(scale * inv_cnt)
                        mx, my = F(s * summ[idx, 0]), F(s * summ[idx, 1])
                        cnt[i] = F(cnt[i] + scale)
                        summ[i, 0] = F(F(mx * scale) + summ[i, 0])
                        summ[i, 1] = F(F(my * scale) + summ[i, 1])
                        added = True
            if not added:                                 # the reference re-inserts the cell and would spin; cannot happen
                raise RuntimeError("interpolate_empty_cells: isolated queue")
            for ox, oy in nbrs:
                nx, ny = x + ox, y + oy
                if 0 <= nx < w and 0 <= ny < h:
                    idx = nx + ny * w
                    key = -calc_counts(idx) + 1
                    if (key, idx) in queue:
                        queue.remove((key, idx)); queue.add((key - 1, idx))
                    elif key != 0 and cnt[idx] < F(0.1):
                        raise AssertionError("unreachable!() of motion_field.rs:288")
    with np.errstate(divide="ignore", invalid="ignore"):
        return (summ / cnt[:, None]).astype(F).reshape(h, w, 2)
