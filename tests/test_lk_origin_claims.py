"""The two arithmetic facts lk.hip's interior-tile column origins rest on, checked in NumPy float32 --
the oracle's own arithmetic -- on random and adversarial inputs:
  1. for unclamped window columns q_k = x + k - r, k = 0..2r, whose sums fq_k = fl(q_k + u) are >= 0, the floors are
     consecutive (floor_k+1 == floor_k + 1 for every k) exactly when floor_2r - floor_0 == 2r;
  2. for fq >= 0, fq - floor(fq) is exact in float32 (so v_fract_f32, which returns the exact fraction, gives the oracle's
     value)."""
import numpy as np

R = 4


def _cases(rng, n):
    x = rng.integers(R, 8192, n).astype(np.int64)
    kinds = rng.integers(0, 6, n)
    u = np.empty(n, np.float32)
    base = rng.integers(-3, 4, n).astype(np.float32)
    tiny = (2.0 ** rng.integers(-30, -8, n)).astype(np.float32)
    u[:] = rng.uniform(-3.5, 3.5, n).astype(np.float32)
    m = kinds == 1; u[m] = base[m] - tiny[m]            # just below an integer: sums that round up to it (or not) as |q| grows
    m = kinds == 2; u[m] = base[m] + tiny[m]            # just above
    m = kinds == 3; u[m] = base[m]                      # integers
    m = kinds == 4                                       # ... and windows that straddle a power of two (the rounding step doubles)
    p2 = (2 ** rng.integers(3, 13, n)).astype(np.int64)
    x[m] = np.maximum(R, p2[m] + rng.integers(-R, R + 1, n)[m] - np.floor(u[m]).astype(np.int64))
    m = kinds == 5; u[m] = (base[m] - np.float32(0.5) * np.spacing(np.float32(x[m] + base[m]))).astype(np.float32)   # half an ulp below: ties
    return x, u


def test_floors_of_unclamped_nonnegative_sums_are_consecutive_iff_the_last_is_2r_above_the_first():
    rng = np.random.default_rng(20260928)
    x, u = _cases(rng, 400_000)
    k = np.arange(2 * R + 1, dtype=np.int64)
    q = (x[:, None] + k[None, :] - R).astype(np.float32)           # exact: small integers
    fq = q + u[:, None]                                            # float32 sums, rounded once: the oracle's expression
    fl = np.floor(fq)
    ok = fq[:, 0] >= 0
    assert ok.sum() > 300_000
    steps = np.diff(fl, axis=1)
    consecutive = (steps == 1).all(axis=1)
    by_ends = (fl[:, -1] - fl[:, 0]) == 2 * R
    np.testing.assert_array_equal(consecutive[ok], by_ends[ok])
    assert set(np.unique(steps[ok]).tolist()) <= {1.0, 2.0}        # never 0: a floor is never repeated
    assert (~consecutive[ok]).sum() > 0                            # the adversarial cases do produce skipped floors


def test_fraction_of_a_nonnegative_float32_is_exact():
    rng = np.random.default_rng(7)
    fq = np.concatenate([rng.uniform(0, 8192, 300_000).astype(np.float32),
                         (rng.integers(0, 8192, 100_000) - (2.0 ** rng.integers(-24, -1, 100_000))).astype(np.float32).clip(0)])
    frac32 = fq - np.floor(fq)                                     # float32 arithmetic
    frac_exact = fq.astype(np.float64) - np.floor(fq.astype(np.float64))
    np.testing.assert_array_equal(frac32.astype(np.float64), frac_exact)
    assert (frac32 < 1).all() and (frac32 >= 0).all()
