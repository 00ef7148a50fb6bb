"""The cheap exact division by a constant of rpt_math.h IS the IEEE operation: checked on the device for
every float they are used on, not argued (rpt_debug_math_sweep compares bit patterns against the compiler's correctly
rounded expansions; those are themselves compared with the host on samples in test_gpu_parity / test_math)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [8e3, 12e2])          # SKY_H_RAY, SKY_H_MIE (k_sky_generate.h; skybox.rs:10-11)
def test_division_by_a_constant_is_the_ieee_quotient_on_its_whole_domain(renderer, c):
    """div_const_nontiny's contract: numerator zero, not finite, or >= 2^-100 in magnitude — every such float, both signs."""
    lo = 0x0d800000                                   # 2^-100
    for first, count in ((lo, 0x80000000 - lo), (0x80000000 + lo, 0x80000000 - lo), (0, 1), (0x80000000, 1)):
        bad, where = renderer.debug_math_sweep(1, first, count, c)
        assert bad == 0, (c, hex(first), bad, hex(where))


def test_the_sweep_sees_a_difference_where_there_is_one(renderer):
    """Below the contract's range the three-step quotient does lose bits (denormal residuals): the checker is not blind."""
    bad, _ = renderer.debug_math_sweep(1, 1, 0x00800000, 8e3)
    assert bad > 0


def test_device_sqrt_equals_the_host_sqrt_on_samples(renderer):
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.random(1 << 20, dtype=np.float32) * 4.0, np.exp(rng.uniform(-80, 80, 1 << 20)).astype(np.float32),
                        np.array([0.0, -0.0, 1e-45, 1e-38, 1.17549435e-38, 3.4e38, np.inf, 1.0, 2.0, 4.0, 0.25], np.float32)])
    got = renderer.debug_math(7, x)
    assert np.array_equal(got.view(np.uint32), np.sqrt(x).view(np.uint32))
