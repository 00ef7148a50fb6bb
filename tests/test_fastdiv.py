"""rpt_fastdiv.h: the traversal's guarded reciprocal division must equal IEEE division bit for bit (CPU build)."""
import numpy as np


def _check(hipmod, x, y):
    fast = hipmod.debug_math_host(9, x, y)
    true = hipmod.debug_math_host(8, x, y)
    nan = np.isnan(fast) & np.isnan(true)
    # a zero quotient may differ in SIGN only for the dividend -0.0 (x = -0, y > 0); intersect_aabb only ever
    # compares these values (intersection.rs:107-117), so the sign of zero is unobservable there
    zero = (fast == 0) & (true == 0)
    ok = nan | zero
    assert np.array_equal(fast.view(np.uint32)[~ok], true.view(np.uint32)[~ok])
    assert not np.any(np.isnan(fast) ^ np.isnan(true))


def test_fast_division_equals_ieee_division(hipmod):
    rng = np.random.default_rng(17)
    n = 4_000_000
    for _ in range(3):
        # divisors like ray-direction components, dividends like (bound - origin)
        y = (rng.uniform(-1, 1, n)).astype(np.float32)
        y[: n // 8] = (rng.uniform(-1, 1, n // 8) * 10.0 ** rng.uniform(-12, 0, n // 8)).astype(np.float32)
        x = (rng.uniform(-50, 50, n) - rng.uniform(-50, 50, n)).astype(np.float32)
        _check(hipmod, x, y)
    # bit-pattern sweep: random mantissas / exponents across and beyond the guards (beyond -> true division path)
    mant = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    mant[:1000] = 0x7FFFFF
    mant[1000:2000] = 0
    ey = rng.integers(127 - 45, 127 + 4, n, dtype=np.uint32)
    y = ((rng.integers(0, 2, n, dtype=np.uint32) << 31) | (ey << 23) | mant).view(np.float32)
    ex = rng.integers(127 - 90, 127 + 45, n, dtype=np.uint32)
    x = ((rng.integers(0, 2, n, dtype=np.uint32) << 31) | (ex << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint32)).view(np.float32)
    _check(hipmod, x, y)


def test_fast_division_special_operands(hipmod):
    f = np.float32
    x = np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 1e-45, 3e38, 1.0, 1.0, 0.0, 5.0], f)
    y = np.array([0.5, 0.5, 0.0, 0.5, 0.5, 0.5, 0.5, 0.5, np.inf, 1e-45, 0.0, np.nan], f)
    _check(hipmod, x, y)
