"""rpt_math.h against correctly-rounded golden vectors (mpmath, tests/golden/math_cr_vectors.npz made by
tools/gen_math_golden.py) and against the platform libm the reference's CPU path would call."""
import os

import numpy as np
import pytest

from conftest import ROOT

OPS = {"sin": 0, "cos": 1, "acos": 2, "exp": 3, "pow": 4, "asin": 5, "atan2": 6}


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "math_cr_vectors.npz"))


@pytest.mark.parametrize("name", sorted(OPS))
def test_shared_math_is_correctly_rounded(oracle, hipmod, golden, name):
    x = golden[name + "_x"]
    y = golden[name + "_y"] if name + "_y" in golden else None
    want = golden[name + "_r"]
    got_gxx = oracle.math(OPS[name], x, y)                    # g++ build (oracle)
    got_clang = hipmod.debug_math_host(OPS[name], x, y)       # clang host build inside librpt_hip.so
    assert np.array_equal(got_gxx.view(np.uint32), got_clang.view(np.uint32)), "host compilers disagree"
    bad = got_gxx.view(np.uint32) != want.view(np.uint32)
    assert bad.sum() == 0, (name, x[bad][:5], got_gxx[bad][:5], want[bad][:5])


def test_special_values(oracle):
    f = np.float32
    nan, inf = f(np.nan), f(np.inf)
    assert np.isnan(oracle.math(0, [nan, inf, -inf])).all() and np.isnan(oracle.math(1, [nan, inf])).all()
    assert np.array_equal(oracle.math(0, [0.0, -0.0]).view(np.uint32), np.array([0.0, -0.0], f).view(np.uint32))
    assert np.isnan(oracle.math(2, [1.0000001, -2.0, nan])).all() and oracle.math(2, [1.0])[0] == 0.0
    assert np.array_equal(oracle.math(3, [-inf, inf, -200.0, 100.0, 0.0]), np.array([0.0, inf, 0.0, inf, 1.0], f))
    p = oracle.math(4, [0.0, 0.0, 1.0, 2.0, -1.0, -8.0, inf, 0.5], [2.2, -1.0, nan, 0.0, 0.5, 3.0, 2.0, inf])
    assert p[0] == 0.0 and p[1] == inf and p[2] == 1.0 and p[3] == 1.0 and np.isnan(p[4]) and p[5] == -512.0
    assert p[6] == inf and p[7] == 0.0
    a = oracle.math(6, [0.0, 0.0, 1.0, -1.0, inf], [1.0, -1.0, 0.0, 0.0, inf])
    assert a[0] == 0.0 and abs(a[1] - np.pi) < 1e-6 and abs(a[2] - np.pi / 2) < 1e-6 and abs(a[3] + np.pi / 2) < 1e-6
    assert abs(a[4] - np.pi / 4) < 1e-6


def test_distance_from_platform_libm(oracle, oracle_libm):
    """The reference's CPU path calls the platform libm (<1 ulp, not correctly rounded). The shared math may differ
    from it only in the last bit and only on a small fraction of arguments."""
    rng = np.random.default_rng(3)
    n = 400_000
    for op, lo, hi, max_frac in [(0, 0, 6.2832, 0.03), (1, 0, 6.2832, 0.03), (2, 0, 1, 0.12), (3, -30, 0, 0.01)]:
        x = rng.uniform(lo, hi, n).astype(np.float32)
        a, b = oracle.math(op, x), oracle_libm.math(op, x)
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
        assert (ulp != 0).mean() <= max_frac


def test_sky_exp_is_within_one_ulp_of_the_correctly_rounded_value(oracle):
    """rpt_math.h exp_sky: the float-only exp of the sky march (84 calls per miss).  Not correctly rounded by design —
    no decision of a path depends on the sky's radiance — but never more than 1 ulp from the correctly rounded expr, equal
    to it on most arguments, monotone where it matters, and with the same special cases."""
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-104, 89, 400_000), rng.uniform(-3, 0, 400_000), -np.exp(rng.uniform(-20, 7, 200_000)),
                        [0.0, -0.0, 1.0, -1.0, 88.9, 89.5, -103.9, -104.5, -87.4, -100.0, 1e-10, -1e-10]]).astype(np.float32)
    fast, exact = oracle.math(10, x), oracle.math(3, x)
    ulp = np.abs(fast.view(np.int32).astype(np.int64) - exact.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, (x[ulp.argmax()], fast[ulp.argmax()], exact[ulp.argmax()])
    assert (ulp != 0).mean() < 0.15
    assert oracle.math(10, np.float32([np.inf]))[0] == np.inf and oracle.math(10, np.float32([-np.inf]))[0] == 0.0
    assert np.isnan(oracle.math(10, np.float32([np.nan]))[0])
    xs = np.sort(rng.uniform(-20, 0, 100_000).astype(np.float32))
    ys = oracle.math(10, xs)
    assert np.all(np.diff(ys) >= -np.spacing(ys[:-1]))        # never decreases by more than its own last bit


def test_exp_sky_specials_without_branches(tmp_path):
    """exp_sky clamps its argument and passes a NaN through with one select (rpt_math.h, round 4) instead of three special-case branches: the two forms
    agree on every 64th float bit pattern and on EVERY pattern near 89, -104, the infinities and the NaNs (tools/exp_sky_check.cpp; with stride 1 — all 2^32
    floats, ~30 s — it reports 0 mismatches as well)."""
    import subprocess
    exe = tmp_path / "exp_sky_check"
    subprocess.run(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-mfma", "-pthread", "-I" + ROOT, "-o", str(exe), os.path.join(ROOT, "tools", "exp_sky_check.cpp")], check=True)
    out = subprocess.run([str(exe), "64"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-500:]


def test_texel_channels_are_divided_by_255_exactly(hipmod):
    """rptm::unorm8 (an atlas texel channel, src/asset.rs:270 `Vec4(r, g, b, 255) / 255.0`): three instructions instead of an IEEE division, the correctly
    rounded quotient for EVERY channel value — the host build here, the device in tests/test_gpu_parity.py."""
    x = np.arange(256, dtype=np.float32)
    got = hipmod.debug_math_host(11, x)
    assert np.array_equal(got.view(np.uint32), (x / np.float32(255.0)).view(np.uint32))
    assert got[0] == 0.0 and not np.signbit(got[0]) and got[255] == 1.0
