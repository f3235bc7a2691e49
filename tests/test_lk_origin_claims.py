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


def test_clamped_last_floor_defeats_the_end_difference_test_and_the_guard_restores_it():
    """lk_origin clamps a floor to [-1, w].  Across a binade the unclamped floors may advance by 2, and a last floor that was
    clamped down to w can make the END DIFFERENCE read 2r although a floor was skipped (w = 1030, r = 4,
    fq_0 = 1022.99995 -> floors 1022, 1023, 1025, ..., 1031, the last clamped to 1030).  lk.hip therefore accepts the
    two-chain verdict only when the clamped last floor is < w (and the first >= 0): with that guard the verdict equals
    "every clamped floor is its predecessor + 1 and no clamp is active" on random and adversarial inputs."""
    w = 1030
    u = np.float32(1022.99995) - np.float32(1018.0)
    k = np.arange(2 * R + 1, dtype=np.int64)
    fq = (np.int64(1022) + k - R).astype(np.float32) + u
    fl = np.floor(fq)
    assert fl[0] == 1022 and (np.diff(fl) == 2).any()              # a floor is skipped across 1024 ...
    cl = np.clip(fl, -1, w)
    assert cl[-1] - cl[0] == 2 * R                                  # ... and the clamped ends hide it
    assert not (cl[0] >= 0 and cl[-1] < w and cl[-1] - cl[0] == 2 * R)   # the guarded test does not

    rng = np.random.default_rng(20260929)
    n = 400_000
    ws = np.concatenate([2 ** rng.integers(4, 12, n // 2) + rng.integers(1, 2 * R + 2, n // 2), rng.integers(2 * R + 2, 4096, n - n // 2)]).astype(np.int64)
    x = np.clip(ws - 1 - R - rng.integers(0, 3 * R, n), R, None)                        # interior tiles: no window COLUMN is clamped
    ok_tile = x + R <= ws - 1
    uu = rng.uniform(-2.0, 12.0, n).astype(np.float32)
    near = rng.random(n) < 0.5                                                          # sums that land on / next to a power of two
    p2 = (2.0 ** np.floor(np.log2(np.maximum(x - R, 2)) + 1)).astype(np.float32)
    uu[near] = (p2[near] - (x[near] - R).astype(np.float32) + rng.integers(-2, 3, near.sum()).astype(np.float32)
                - (2.0 ** rng.integers(-16, -8, near.sum())).astype(np.float32))
    q = (x[:, None] + k[None, :] - R).astype(np.float32)
    fl = np.floor(q + uu[:, None])
    cl = np.clip(fl, -1, ws[:, None].astype(np.float32))
    guarded = (cl[:, 0] >= 0) & (cl[:, -1] < ws) & (cl[:, -1] - cl[:, 0] == 2 * R)
    truth = (np.diff(cl, axis=1) == 1).all(axis=1) & (cl[:, 0] >= 0) & (cl[:, -1] < ws)
    np.testing.assert_array_equal(guarded[ok_tile], truth[ok_tile])
    unguarded = (cl[:, 0] >= 0) & (cl[:, -1] - cl[:, 0] == 2 * R)
    wrong = unguarded & ~(np.diff(cl, axis=1) == 1).all(axis=1) & ok_tile
    assert wrong.sum() > 0                                          # the random set does contain the round-3 bug's inputs
