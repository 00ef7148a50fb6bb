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


def test_float_to_int_conversion_is_rusts_as_i32_for_every_float(renderer):
    """rptm::f2i32_sat on the device is ONE v_cvt_i32_f32 (image_polyfill.rs:41-42 `as_ivec2`: 16 conversions per textured hit): equal to the written-out
    saturating cast — NaN -> 0, truncation, i32::MIN / MAX beyond the range — on all 2^32 bit patterns."""
    for first in (0, 0x80000000):
        bad, where = renderer.debug_math_sweep(2, first, 0x80000000, 1.0)
        assert bad == 0, (hex(first), bad, hex(where))


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


def test_device_exp_sky_at_its_special_arguments(renderer, hipmod):
    """exp_sky without special-case branches (rpt_math.h): on the DEVICE the clamp is __builtin_fminf / fmaxf, another expression than
    the host's — known answers at NaN, the infinities, the clamp bounds and their neighbours, and device == host build bit for bit
    on windows of every float around them."""
    nan = np.float32(np.nan)
    x = np.array([nan, -nan, np.inf, -np.inf, 89.0, np.nextafter(np.float32(89.0), np.float32(np.inf)), np.nextafter(np.float32(89.0), np.float32(0)),
                  -104.0, np.nextafter(np.float32(-104.0), np.float32(-np.inf)), np.nextafter(np.float32(-104.0), np.float32(0)), 0.0, -0.0, 3.0e38, -3.0e38,
                  88.5, -103.0], np.float32)
    got = renderer.debug_math(10, x)
    assert np.isnan(got[0]) and np.isnan(got[1])
    assert got[2] == np.inf and got[3] == 0.0 and not np.signbit(got[3])
    assert got[4] == np.inf and got[5] == np.inf and got[6] == np.inf          # e^88.72... is the largest float already
    assert got[7] == 0.0 and got[8] == 0.0 and got[9] == 0.0 and got[10] == 1.0 and got[11] == 1.0
    assert got[12] == np.inf and got[13] == 0.0
    assert abs(float(got[14]) / 2.723088e38 - 1.0) < 1e-6 and got[15].view(np.uint32) == 1      # a large finite value; the smallest denormal
    for centre in (np.float32(89.0), np.float32(-104.0), np.float32(np.inf), np.float32(-np.inf), np.float32(88.72), np.float32(-87.3), np.float32(0.0)):
        c = int(np.array([centre], np.float32).view(np.uint32)[0])
        bits = (np.arange(-4096, 4096, dtype=np.int64) + c) & 0xffffffff
        w = bits.astype(np.uint32).view(np.float32)
        dev, host = renderer.debug_math(10, w), hipmod.debug_math_host(10, w)
        both_nan = np.isnan(dev) & np.isnan(host)
        assert np.array_equal(dev.view(np.uint32)[~both_nan], host.view(np.uint32)[~both_nan]), float(centre)
        assert np.array_equal(np.isnan(dev), np.isnan(host))
