"""The reference's Almeida known-answer cases, rebuilt from the oracle's primitives.

Follows almeida-estimator/src/lib.rs:253-358 (get_grid, calc_view, project_grid, calc_field,
test_rot): a 50x50 grid of screen points un-projected with camera (1.0, 90 deg), re-projected
under 4 magnitudes x 8 Euler combinations; the field keeps a point when either endpoint is
within 0.71 of the centre.  The acceptance bound is `angle_to(true, est) < 0.1 * rot` degrees
(lib.rs:343-348).
"""
import numpy as np

import oracle

ROTS = [0.01, 0.1, 1.0, 10.0]                                           # lib.rs:313


def angle_combos(rot):                                                  # lib.rs:314-323
    return [(0.0, 0.0, 0.0), (rot, 0.0, 0.0), (0.0, rot, 0.0), (0.0, 0.0, rot),
            (rot, rot, 0.0), (rot, 0.0, rot), (0.0, rot, rot), (rot, rot, rot)]


def calc_view(q):                                                       # lib.rs:280-286
    pos = np.zeros(3, np.float32)
    fwd = oracle.quat_transform_vector(q, [0.0, -1.0, 0.0])
    up = oracle.quat_transform_vector(q, [0.0, 0.0, 1.0])
    return oracle.look_at_rh(pos, pos + fwd, up)


def get_grid(cam, nx=50, ny=50):                                        # lib.rs:257-278
    ident = np.array([1, 0, 0, 0], np.float32)
    view = calc_view(ident)
    pts = []
    for x in range(nx):
        for y in range(ny):
            p = np.array([np.float32(x) / np.float32(nx), np.float32(y) / np.float32(ny)], np.float32)
            pts.append(oracle.camera_unproject(cam, p, view))
    return np.array(pts, np.float32)


def project_grid(cam, grid, view):                                      # lib.rs:288-294
    return np.array([oracle.camera_project(cam, p, view) for p in grid], np.float32)


def calc_field(p1, p2):                                                 # lib.rs:296-306
    mid = np.array([0.5, 0.5], np.float32)
    d1 = (p1 - mid).astype(np.float32); d2 = (p2 - mid).astype(np.float32)
    m1 = np.sqrt((d1[:, 0] * d1[:, 0] + d1[:, 1] * d1[:, 1]).astype(np.float32))
    m2 = np.sqrt((d2[:, 0] * d2[:, 0] + d2[:, 1] * d2[:, 1]).astype(np.float32))
    keep = (m1 <= np.float32(0.71)) | (m2 <= np.float32(0.71))
    return np.concatenate([p1[keep], (p2 - p1).astype(np.float32)[keep]], axis=1).astype(np.float32)


_CACHE = {}


def cases():
    """-> list of (rot_deg, (r,p,y) degrees, true quaternion wijk, entries[N,4])"""
    if "cases" in _CACHE:
        return _CACHE["cases"]
    cam = oracle.camera(1.0, 90.0)                                      # lib.rs:309
    grid = get_grid(cam)
    ident = np.array([1, 0, 0, 0], np.float32)
    p1 = project_grid(cam, grid, calc_view(ident))
    out = []
    k = np.float32(np.pi) / np.float32(180.0)
    for rot in ROTS:
        for (r, p, y) in angle_combos(rot):
            q = oracle.quat_from_euler(np.float32(r) * k, np.float32(p) * k, np.float32(y) * k)
            p2 = project_grid(cam, grid, calc_view(q))
            out.append((rot, (r, p, y), q, calc_field(p1, p2)))
    _CACHE["cases"] = out
    return out


def error_deg(q_true, q_est) -> float:
    return float(np.degrees(oracle.quat_angle_to(q_true, q_est)))
