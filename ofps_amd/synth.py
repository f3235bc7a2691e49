"""Seeded synthetic inputs for the flow hot path (SURVEY.md 8d "Synthetic inputs").

Everything is a pure function of (seed, shape) built from a counter-based integer hash, so
the CPU baseline, the parity tests and the GPU bench see identical bytes.

* `luma_sequence`: F luma frames of one endless multi-octave value-noise texture; every
  64x64 region of every frame samples the texture at its own integer offset, and the offset
  moves by an integer step in [-max_step, max_step]^2 per frame, so each pair (k, k+1) holds
  planted integer block displacements; +-1 uniform noise on top.
* `rotation_field`: MotionEntry records ((x+.5)/W, (y+.5)/H, delta) as cv-decoder emits them
  (cv-decoder/src/lib.rs:262-269) for a planted camera rotation plus Gaussian noise.
"""
from __future__ import annotations

import numpy as np

SEED0 = 0x0F950001


def _hash_u32(x: np.ndarray) -> np.ndarray:
    """lowbias32-style avalanche on uint32 arrays."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _lattice(ix: np.ndarray, iy: np.ndarray, salt: int) -> np.ndarray:
    """float32 in [0,1) per integer lattice point."""
    with np.errstate(over="ignore"):
        h = _hash_u32(ix.astype(np.uint32) * np.uint32(0x9E3779B1)
                      ^ _hash_u32(iy.astype(np.uint32) * np.uint32(0x85EBCA77) + np.uint32(salt & 0xFFFFFFFF)))
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def _value_noise(xs: np.ndarray, ys: np.ndarray, period: int, salt: int) -> np.ndarray:
    """Bilinear value noise at integer sample coordinates xs, ys (int64 arrays, same shape)."""
    ix = np.floor_divide(xs, period); iy = np.floor_divide(ys, period)
    fx = ((xs - ix * period).astype(np.float32) + np.float32(0.5)) / np.float32(period)
    fy = ((ys - iy * period).astype(np.float32) + np.float32(0.5)) / np.float32(period)
    v00 = _lattice(ix, iy, salt); v10 = _lattice(ix + 1, iy, salt)
    v01 = _lattice(ix, iy + 1, salt); v11 = _lattice(ix + 1, iy + 1, salt)
    top = v00 + (v10 - v00) * fx
    bot = v01 + (v11 - v01) * fx
    return top + (bot - top) * fy


def _value_noise_grid(x1: np.ndarray, y1: np.ndarray, period: int, salt: int) -> np.ndarray:
    """_value_noise on the mesh y1 (rows) x x1 (columns) of int64 coordinates: the lattice is hashed once per lattice point
    instead of four times per pixel; same f32 operations per element, same bits."""
    ix = np.floor_divide(x1, period); iy = np.floor_divide(y1, period)
    fx = (((x1 - ix * period).astype(np.float32) + np.float32(0.5)) / np.float32(period))[None, :]
    fy = (((y1 - iy * period).astype(np.float32) + np.float32(0.5)) / np.float32(period))[:, None]
    ux = np.arange(ix.min(), ix.max() + 2, dtype=np.int64); uy = np.arange(iy.min(), iy.max() + 2, dtype=np.int64)
    lat = _lattice(np.broadcast_to(ux[None, :], (len(uy), len(ux))), np.broadcast_to(uy[:, None], (len(uy), len(ux))), salt)
    cx = (ix - ux[0]).astype(np.intp); cy = (iy - uy[0]).astype(np.intp)
    rows0 = lat[cy]; rows1 = lat[cy + 1]                       # [h, lattice columns]
    v00 = rows0[:, cx]; v10 = rows0[:, cx + 1]; v01 = rows1[:, cx]; v11 = rows1[:, cx + 1]
    top = v00 + (v10 - v00) * fx
    bot = v01 + (v11 - v01) * fx
    return top + (bot - top) * fy


def _texture_grid(x1: np.ndarray, y1: np.ndarray, seed: int) -> np.ndarray:
    """_texture on a mesh (rows y1, columns x1)."""
    return (np.float32(4) * _value_noise_grid(x1, y1, 64, seed) + np.float32(2) * _value_noise_grid(x1, y1, 16, seed + 1)
            + _value_noise_grid(x1, y1, 4, seed + 2)) / np.float32(7)


def _texture(xs: np.ndarray, ys: np.ndarray, seed: int) -> np.ndarray:
    """Octaves at 64/16/4 px, weights 4:2:1 -> float32 in [0,1)."""
    n = (np.float32(4) * _value_noise(xs, ys, 64, seed) + np.float32(2) * _value_noise(xs, ys, 16, seed + 1)
         + _value_noise(xs, ys, 4, seed + 2)) / np.float32(7)
    return n


def luma_sequence(n_frames: int, width: int, height: int, max_step: int, seed: int = SEED0,
                  region: int = 64, noise: int = 1, stride: int | None = None) -> np.ndarray:
    """-> uint8 [n_frames, height, stride] (stride >= width, padding zero)."""
    stride = width if stride is None else stride
    assert stride >= width
    rx = (width + region - 1) // region; ry = (height + region - 1) // region
    yy, xx = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    reg_id = (yy // region) * rx + (xx // region)
    ridx = np.arange(rx * ry, dtype=np.uint32)
    out = np.zeros((n_frames, height, stride), np.uint8)
    span = 2 * max_step + 1
    # per-frame region offsets first, then ONE evaluation of the texture over the bounding box of every coordinate
    # any frame samples (the texture is a pure function of integer coordinates, so gathering from that canvas gives
    # the same bytes as evaluating it per frame, at a fraction of the cost)
    offs = np.zeros((n_frames, rx * ry, 2), np.int64)
    for k in range(1, n_frames):
        with np.errstate(over="ignore"):
            hx = _hash_u32(ridx * np.uint32(2654435761) + np.uint32((seed + 101 * k) & 0xFFFFFFFF))
            hy = _hash_u32(hx + np.uint32(0x68E31DA4))
        offs[k, :, 0] = offs[k - 1, :, 0] + (hx % np.uint32(span)).astype(np.int64) - max_step
        offs[k, :, 1] = offs[k - 1, :, 1] + (hy % np.uint32(span)).astype(np.int64) - max_step
    lo_x, hi_x = int(offs[..., 0].min()), int(offs[..., 0].max())
    lo_y, hi_y = int(offs[..., 1].min()), int(offs[..., 1].max())
    cw, ch = width + hi_x - lo_x, height + hi_y - lo_y
    if cw * ch <= 64 * 1024 * 1024:
        canvas = _texture_grid(np.arange(cw, dtype=np.int64) + lo_x + 100000, np.arange(ch, dtype=np.int64) + lo_y + 100000, seed)
    else:
        canvas = None                                     # very long sequences: fall back to per-frame evaluation
    for k in range(n_frames):
        off = offs[k]
        if canvas is not None:
            tex = canvas[yy + off[reg_id, 1] - lo_y, xx + off[reg_id, 0] - lo_x]
        else:
            tex = _texture(xx + off[reg_id, 0] + 100000, yy + off[reg_id, 1] + 100000, seed)
        img = np.float32(16.0) + tex * np.float32(219.0)
        if noise:
            with np.errstate(over="ignore"):
                hn = _hash_u32((yy * width + xx).astype(np.uint32) * np.uint32(0xC2B2AE35)
                               + np.uint32((seed + 7777 * (k + 1)) & 0xFFFFFFFF))
            img = img + ((hn % np.uint32(2 * noise + 1)).astype(np.float32) - np.float32(noise))
        out[k, :, :width] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def flatten_regions(frames: np.ndarray, region: int = 40, seed: int = SEED0, keep: int = 2) -> np.ndarray:
    """Copy of `frames` with about (keep-1)/keep of the region x region squares painted a flat grey: low-contrast
    areas for cv-decoder's Sobel mask (cv-decoder/src/lib.rs:203-237) to reject."""
    out = np.array(frames, np.uint8, copy=True)
    h, w = out.shape[-2:]
    yy, xx = np.meshgrid(np.arange(h, dtype=np.uint32), np.arange(w, dtype=np.uint32), indexing="ij")
    with np.errstate(over="ignore"):
        hid = _hash_u32((yy // np.uint32(region)) * np.uint32(7919) + (xx // np.uint32(region)) + np.uint32(seed & 0xFFFFFFFF))
    flat = (hid % np.uint32(keep)) != 0
    out[..., flat] = 128
    return out


def random_luma(n_frames: int, width: int, height: int, seed: int = SEED0) -> np.ndarray:
    """Unstructured uint8 noise frames (worst case for SAD ties: many near-equal costs)."""
    idx = np.arange(n_frames * height * width, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = _hash_u32(idx * np.uint32(0x9E3779B1) + np.uint32(seed & 0xFFFFFFFF))
    return (h >> np.uint32(24)).astype(np.uint8).reshape(n_frames, height, width)


def _gauss(n: int, seed: int) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        u1 = (_hash_u32(idx * np.uint32(0x9E3779B1) + np.uint32(seed & 0xFFFFFFFF)) >> np.uint32(8)).astype(np.float64)
        u2 = (_hash_u32(idx * np.uint32(0x85EBCA77) + np.uint32((seed + 1) & 0xFFFFFFFF)) >> np.uint32(8)).astype(np.float64)
    u1 = (u1 + 1.0) / float(1 << 24); u2 = u2 / float(1 << 24)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def rotation_delta(pos: np.ndarray, aspect: float, fov_y_deg: float, rot3: np.ndarray) -> np.ndarray:
    """Screen-space displacement of normalised points under a camera rotation
    (the model of ofps/src/camera.rs:89-117 in float64 closed form; generator only)."""
    pos = np.asarray(pos, np.float64).reshape(-1, 2)
    t = np.tan(np.radians(fov_y_deg) / 2.0)
    cx = pos[:, 0] * 2 - 1; cy = pos[:, 1] * 2 - 1
    w = np.stack([-cx * t * aspect, -np.ones_like(cx), cy * t], 1)
    r = w @ np.asarray(rot3, np.float64).T
    sx = (-r[:, 0]) / (-r[:, 1]) / (t * aspect)
    sy = r[:, 2] / (-r[:, 1]) / t
    out = np.stack([(sx + 1) * 0.5, (sy + 1) * 0.5], 1) - pos
    return out


def euler_rot3(roll: float, pitch: float, yaw: float) -> np.ndarray:
    sr, cr = np.sin(roll), np.cos(roll); sp, cp = np.sin(pitch), np.cos(pitch); sy, cy = np.sin(yaw), np.cos(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def rotation_field(width: int, height: int, aspect: float = 16.0 / 9.0, fov_y_deg: float = 39.6 * 9.0 / 16.0,
                   euler_deg=(0.5, 0.3, -0.2), sigma: float = 2e-4, seed: int = SEED0 + 3,
                   outlier_frac: float = 0.0) -> np.ndarray:
    """-> float32 [width*height, 4] MotionEntry records in raster order (cfg3 input)."""
    yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    pos = np.stack([(xx.ravel() + 0.5) / width, (yy.ravel() + 0.5) / height], 1)
    r = euler_rot3(*np.radians(euler_deg))
    d = rotation_delta(pos, aspect, fov_y_deg, r)
    n = pos.shape[0]
    d[:, 0] += sigma * _gauss(n, seed); d[:, 1] += sigma * _gauss(n, seed + 17)
    if outlier_frac > 0:                      # a deterministic subset becomes gross outliers
        k = int(n * outlier_frac)
        idx = np.argsort(_gauss(n, seed + 99), kind="stable")[:k]
        d[idx] = 0.02 * np.stack([_gauss(k, seed + 5), _gauss(k, seed + 6)], 1)
    return np.concatenate([pos, d], 1).astype(np.float32)


def camera_warp_pair(W=1920, H=1080, roll_deg=0.5, zoom=1.004, shift=(3.3, -2.6), seed=11):
    """A pair related by a SMOOTH sub-pixel motion field (small roll + zoom + shift about the image centre: up to ~+-12 px at the
    corners of a 1080p frame, no discontinuities): the kind of flow a moving camera produces, against the region-wise integer
    jumps of synth.luma_sequence.  The second frame is the first one's texture sampled bilinearly at the displaced positions."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    tex = ndimage.gaussian_filter(rng.uniform(0, 255, (H + 64, W + 64)).astype(np.float32), 1.2)
    tex = (tex - tex.min()) / (tex.max() - tex.min()) * 255.0
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cx, cy, a = (W - 1) / 2, (H - 1) / 2, np.deg2rad(roll_deg)
    dx, dy = xx - cx, yy - cy
    sx = cx + zoom * (np.cos(a) * dx - np.sin(a) * dy) + shift[0]
    sy = cy + zoom * (np.sin(a) * dx + np.cos(a) * dy) + shift[1]
    f0 = tex[32:32 + H, 32:32 + W]
    f1 = ndimage.map_coordinates(tex, [sy + 32, sx + 32], order=1, mode="nearest")
    noise = rng.integers(-1, 2, (2, H, W))
    return np.clip(np.rint(np.stack([f0, f1])) + noise, 0, 255).astype(np.uint8)


def rotation_clip(per_frame_euler_deg, W=1920, H=1080, fov_y_deg=60.0, seed=21, distractor=None, margin=0.5, noise=1):
    """A clip rendered from planted per-frame CAMERA ROTATIONS (what the reference evaluates against:
    ground-truth camera orientations per frame, ofps-suite/src/app/tracking/mod.rs:125-217).

    per_frame_euler_deg: [n, 3] (roll, pitch, yaw) in nalgebra's from_euler_angles order, the rotation q_k between frame k and
    frame k + 1 in the convention of the reference's own estimator test (almeida-estimator/src/lib.rs:280-306: frame k + 1 is
    frame k seen through calc_view(q_k)), i.e. the quaternion Estimator::estimate should return for the pair.  In image terms, with
    the reference's Z-up / Y-forward world: roll = about X = tilt, pitch = about Y (the view axis) = image roll, yaw = about Z = pan.
    The scene is a texture at infinity (pure rotation has no parallax): frame k samples it, bilinearly, where the pinhole
    homography of the accumulated rotation Q_k = q_0 q_1 ... q_{k-1} puts each pixel centre -- float64, no small-angle
    approximation, no NDC-z quirk (a physical camera, not the estimator's model).
    distractor: None | dict(size=(w, h) px, start=(x, y) px, velocity=(vx, vy) px per frame): an independently moving foreground
    rectangle with its own texture pasted over every frame (a dynamic object: the reference's "dyn" clips).
    -> (frames uint8 [n + 1, H, W], quats float64 [n, 4] (w, i, j, k) = the q_k)."""
    from scipy import ndimage
    from scipy.spatial.transform import Rotation
    eul = np.radians(np.asarray(per_frame_euler_deg, np.float64).reshape(-1, 3))
    n = len(eul)
    rng = np.random.default_rng(seed)
    mw, mh = int(round(W * margin)), int(round(H * margin))
    TW, TH = W + 2 * mw, H + 2 * mh
    u = rng.uniform(0, 1, (TH, TW)).astype(np.float32)
    tex = ndimage.gaussian_filter(u, 1.2) * 3.0 + ndimage.gaussian_filter(u, 4.0) * 6.0 + ndimage.gaussian_filter(u, 16.0) * 12.0
    tex = (tex - tex.min()) / (tex.max() - tex.min()) * 219.0 + 16.0
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pos = np.stack([(xx.ravel() + 0.5) / W, (yy.ravel() + 0.5) / H], 1)
    aspect = W / H
    if distractor:
        dw, dh = distractor["size"]
        du = rng.uniform(0, 1, (dh, dw)).astype(np.float32)
        dtex = ndimage.gaussian_filter(du, 1.0) * 3.0 + ndimage.gaussian_filter(du, 3.0) * 6.0
        dtex = (dtex - dtex.min()) / (dtex.max() - dtex.min()) * 219.0 + 16.0
    frames = np.empty((n + 1, H, W), np.uint8)
    Q = np.eye(3)
    quats = np.empty((n, 4))
    for k in range(n + 1):
        p0 = pos + rotation_delta(pos, aspect, fov_y_deg, Q)                         # M(Q_k)^-1: where this frame's pixels sit in frame 0
        sx = p0[:, 0] * W - 0.5 + mw
        sy = p0[:, 1] * H - 0.5 + mh
        if sx.min() < 0 or sy.min() < 0 or sx.max() > TW - 1 or sy.max() > TH - 1:
            raise ValueError(f"rotation_clip: frame {k} looks outside the texture (accumulated rotation too large for margin {margin})")
        img = ndimage.map_coordinates(tex, [sy.reshape(H, W), sx.reshape(H, W)], order=1, mode="nearest")
        if distractor:
            x0 = int(round(distractor["start"][0] + k * distractor["velocity"][0]))
            y0 = int(round(distractor["start"][1] + k * distractor["velocity"][1]))
            xa, ya, xb, yb = max(x0, 0), max(y0, 0), min(x0 + dw, W), min(y0 + dh, H)
            if xb > xa and yb > ya:
                img[ya:yb, xa:xb] = dtex[ya - y0:yb - y0, xa - x0:xb - x0]
        if noise:
            img = img + rng.integers(-noise, noise + 1, (H, W))
        frames[k] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if k < n:
            Rq = euler_rot3(*eul[k])
            x, y, z, w = Rotation.from_matrix(Rq).as_quat()
            quats[k] = (w, x, y, z)
            Q = Q @ Rq                                                            # Q_{k+1} = Q_k q_k
    return frames, quats


def quat_angle_deg(a, b) -> float:
    """Angle of the rotation that takes a to b (UnitQuaternion::angle_to), degrees; (w, i, j, k).  From the VECTOR part of
    a^-1 b (atan2), not from arccos of the dot product: for rotations of hundredths of a degree 1 - dot is ~1e-9."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    a = a / np.linalg.norm(a); b = b / np.linalg.norm(b)
    aw, av = a[0], -a[1:]                                   # conj(a)
    w = aw * b[0] - float(np.dot(av, b[1:]))
    v = aw * b[1:] + b[0] * av + np.cross(av, b[1:])
    return float(np.degrees(2.0 * np.arctan2(np.linalg.norm(v), abs(w))))
