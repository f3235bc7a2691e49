"""An INDEPENDENT float64 model of the reference's camera and test geometry, written from the Rust text
(ofps/src/camera.rs:26-117, almeida-estimator/src/lib.rs:257-306) with generic 4x4 linear algebra: textbook OpenGL
perspective matrix, numpy.linalg.inv for every inverse, scipy's Rotation for Euler angles and quaternions.  It imports
nothing from oracle/ or ofps_amd/, so an error shared by the oracle's hand-simplified closed forms (an axis, a sign,
the NDC-z divide) cannot cancel out in tests that compare the two."""
import numpy as np
from scipy.spatial.transform import Rotation

VIEW = np.array([[-1.0, 0, 0, 0], [0, 0, 1.0, 0], [0, 1.0, 0, 0], [0, 0, 0, 1.0]])       # camera.rs:91-96


def perspective(aspect, fovy_rad, zn=0.1, zf=10.0):
    """nalgebra Perspective3::new == the OpenGL gluPerspective matrix (camera.rs:27)."""
    f = 1.0 / np.tan(fovy_rad / 2.0)
    return np.array([[f / aspect, 0, 0, 0], [0, f, 0, 0], [0, 0, (zf + zn) / (zn - zf), 2 * zf * zn / (zn - zf)], [0, 0, -1.0, 0]])


def transform_point(M, p):
    """Matrix4::transform_point: homogeneous multiply with w = 1, divide by the resulting w."""
    p = np.atleast_2d(np.asarray(p, np.float64))
    h = np.concatenate([p, np.ones((len(p), 1))], 1) @ M.T
    return h[:, :3] / h[:, 3:4]


def unproject(P, screen, inv_view):
    """camera.rs:45-55."""
    c = np.atleast_2d(np.asarray(screen, np.float64)) * 2.0 - 1.0
    return transform_point(inv_view @ np.linalg.inv(P), np.concatenate([c, np.ones((len(c), 1))], 1))


def project(P, world, view, ndc_z_divide=True):
    """camera.rs:72-81: Perspective3::project_point gives NDC (x, y, z); the reference then divides x and y by that NDC z
    (:77) before mapping to [0,1]."""
    v = transform_point(view, world)
    h = np.concatenate([v, np.ones((len(v), 1))], 1) @ P.T
    ndc = h[:, :3] / h[:, 3:4]
    xy = ndc[:, :2] / ndc[:, 2:3] if ndc_z_divide else ndc[:, :2]
    return (xy + 1.0) * 0.5


def euler_rot3(roll, pitch, yaw):
    """nalgebra from_euler_angles(roll, pitch, yaw) = Rz(yaw) Ry(pitch) Rx(roll) = scipy extrinsic 'xyz'."""
    return Rotation.from_euler("xyz", [roll, pitch, yaw]).as_matrix()


def rot4(R3):
    M = np.eye(4); M[:3, :3] = R3
    return M


def delta(pos, aspect, fov_y_deg, R3, ndc_z_divide=True):
    """camera.rs:89-117: rotate(coords, R) - coords with the fixed Z-up / Y-forward view."""
    P = perspective(aspect, np.radians(fov_y_deg))
    pos = np.atleast_2d(np.asarray(pos, np.float64))
    w = unproject(P, pos, VIEW.T)
    w = transform_point(rot4(R3), w)
    return project(P, w, VIEW, ndc_z_divide) - pos


def look_at_rh(eye, target, up):
    f = target - eye; f = f / np.linalg.norm(f)
    s = np.cross(f, up); s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    M = np.eye(4)
    M[0, :3], M[1, :3], M[2, :3] = s, u, -f
    M[:3, 3] = -M[:3, :3] @ eye
    return M


def calc_view(rot: Rotation):
    """almeida-estimator/src/lib.rs:280-286."""
    pos = np.zeros(3)
    return look_at_rh(pos, pos + rot.apply([0.0, -1.0, 0.0]), rot.apply([0.0, 0.0, 1.0]))


def almeida_test_field(roll_deg, pitch_deg, yaw_deg, aspect=1.0, fov_y_deg=90.0, n=50):
    """-> (true rotation as scipy Rotation, entries[N,4] float64): get_grid / project_grid / calc_field of
    almeida-estimator/src/lib.rs:257-306 (the test passes the identity VIEW matrix as `inv_view`, :274)."""
    P = perspective(aspect, np.radians(fov_y_deg))
    xs, ys = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")                     # x outer, y inner (:264-270)
    scr = np.stack([xs.ravel() / n, ys.ravel() / n], 1)
    ident = Rotation.identity()
    grid = unproject(P, scr, calc_view(ident))
    q = Rotation.from_euler("xyz", np.radians([roll_deg, pitch_deg, yaw_deg]))
    p1 = project(P, grid, calc_view(ident))
    p2 = project(P, grid, calc_view(q))
    keep = (np.linalg.norm(p1 - 0.5, axis=1) <= 0.71) | (np.linalg.norm(p2 - 0.5, axis=1) <= 0.71)
    return q, np.concatenate([p1, p2 - p1], 1), keep


def quat_wijk(rot: Rotation):
    x, y, z, w = rot.as_quat()
    return np.array([w, x, y, z])
