"""Property tests (hypothesis) of the CPU oracle against its second restatement on adversarial inputs: positions outside
[0,1], NaN / infinities, huge motions, degenerate grids -- the corners where a restatement of nalgebra's `clamp`, Rust's
`round` and saturating `as usize` casts is most likely to go wrong."""
import numpy as np
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

import oracle
from oracle import np_oracle as npo

F32 = dict(dtype=np.float32)
coord = st.one_of(st.floats(-2, 3, width=32), st.sampled_from([0.0, 1.0, 0.5, -0.0, float("nan"), float("inf"), -float("inf"), 1e-30, 1 - 1e-7]))
motion = st.one_of(st.floats(-10, 10, width=32), st.sampled_from([0.0, 1e-8, -1e-8, 3.0e38, -3.0e38]))


@st.composite
def entries(draw, max_n=60):
    n = draw(st.integers(0, max_n))
    e = np.empty((n, 4), np.float32)
    for i in range(n):
        e[i] = (draw(coord), draw(coord), draw(motion), draw(motion))
    return e


@settings(max_examples=120, deadline=None, derandomize=True)
@given(entries(), st.integers(1, 20), st.integers(1, 20))
def test_densify_cells_and_sums_agree_on_adversarial_inputs(e, w, h):
    with np.errstate(all="ignore"):
        f_c, cells_c = oracle.densify(e, w, h, want_cells=True)
        x, y = npo.cell_index(e, w, h) if len(e) else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        f_n = npo.densify(e, w, h)[0] if len(e) else None
    assert (cells_c[: len(e), 0] == x).all() and (cells_c[: len(e), 1] == y).all()
    assert (x < w).all() and (y < h).all()
    if f_n is not None:
        np.testing.assert_array_equal(f_c.view(np.uint32), np.asarray(f_n, np.float32).view(np.uint32))


@settings(max_examples=60, deadline=None, derandomize=True)
@given(entries(max_n=120), st.sampled_from([(0.05, 3), (0.2, 2), (0.5, 1), (0.01, 4)]), st.sampled_from([0.0001, 0.003, 0.05, 0.1]))
def test_detect_motion_agrees_on_adversarial_inputs(e, geom, target):
    e = np.nan_to_num(e, nan=0.25, posinf=2.0, neginf=-1.0)          # the detector's sqrt(x^2+y^2) >= t on NaN is its own topic
    e[:, 2:] = np.clip(e[:, 2:], -1e3, 1e3)
    a = oracle.detect_motion(e, geom[0], geom[1], target)
    b = npo.detect_motion(e, geom[0], geom[1], target)
    assert (a is None) == (b is None)
    if a is not None:
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1].view(np.uint32), np.asarray(b[1], np.float32).view(np.uint32))
