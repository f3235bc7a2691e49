"""-m gpu property tests (hypothesis): the HIP densifier / detector against the CPU oracle on adversarial inputs --
positions outside [0,1], NaN / infinities, huge motions, degenerate grids."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from test_oracle_properties import entries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ofps_amd.runtime import HipContext
    c = HipContext(0)
    yield c
    c.close()


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(e=entries(max_n=200), w=st.integers(1, 40), h=st.integers(1, 40))
def test_hip_densify_matches_oracle_on_adversarial_inputs(ctx, e, w, h):
    f_g, cells_g = ctx.densify(e, w, h, want_cells=True)
    f_o, cells_o = oracle.densify(e, w, h, want_cells=True)
    np.testing.assert_array_equal(cells_g[: len(e)], cells_o[: len(e)])
    np.testing.assert_array_equal(f_g.view(np.uint32), f_o.view(np.uint32))        # NaN payloads included


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(e=entries(max_n=300), geom=st.sampled_from([(0.05, 3), (0.2, 2), (0.5, 1), (0.01, 4)]), target=st.sampled_from([0.0001, 0.003, 0.05]))
def test_hip_detect_matches_oracle_on_adversarial_inputs(ctx, e, geom, target):
    e = np.nan_to_num(e, nan=0.25, posinf=2.0, neginf=-1.0)
    e[:, 2:] = np.clip(e[:, 2:], -1e3, 1e3)
    a = ctx.detect(e, geom[0], geom[1], target)
    b = oracle.detect_motion(e, geom[0], geom[1], target)
    assert (a is None) == (b is None)
    if a is not None:
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
