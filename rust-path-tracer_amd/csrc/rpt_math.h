/*
 * rpt_math.h — deterministic float transcendentals shared by the gfx950 kernels
 * and the CPU oracle.
 *
 * Why this exists: the reference's hot path (the kernels crate) calls
 * f32::{sin,cos,acos,asin,atan2,exp,powf}, which on the CPU path resolve to the
 * platform libm.  A path tracer compares such values against
 * random numbers (lobe choice kernels/src/bsdf.rs:282, light pick
 * kernels/src/light_pick.rs:10-11, roulette kernels/src/lib.rs:177), so CPU/GPU
 * radiance parity needs these functions to be BIT-identical on both sides.
 *
 * How: every function converts its float argument to double and evaluates in
 * double using only IEEE-754 add / mul / fma / compare / convert (no division,
 * no sqrt, no table, no libm call), then rounds once to float.  Those
 * operations are exactly specified on x86-64 and on gfx950 (v_fma_f64,
 * v_add_f64, v_mul_f64, v_cvt_*), so the two sides agree bit for bit provided
 * both are compiled with -ffp-contract=off (explicit fma only).
 * Internal error is ~1e-16 relative, so the float result is the correctly
 * rounded value except when the exact result lies within ~1e-8 ulp of a
 * rounding boundary — i.e. these are "correctly rounded libm" stand-ins, and
 * differ from glibc's <1-ulp float routines in well under 1 % of arguments
 * (measured in tests/test_math.py).
 *
 * Constants come from tools/gen_math_consts.py (mpmath, 120 digits).
 */
#ifndef RPT_MATH_H
#define RPT_MATH_H

#include <stdint.h>
#include "rpt_math_consts.h"

#if defined(__HIPCC__)
#define RPT_HD __host__ __device__ __forceinline__
#else
#define RPT_HD inline __attribute__((always_inline))
#endif
#if defined(__clang__)
#define RPT_UNROLL _Pragma("unroll")
#else
#define RPT_UNROLL _Pragma("GCC unroll 32")
#endif

namespace rptm {

#define RPT_COEF(name, NAME) const double c[NAME##_N] = NAME##_INIT

/* ---- bit casts ---------------------------------------------------------- */
RPT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RPT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RPT_HD uint64_t d2u(double f) { return __builtin_bit_cast(uint64_t, f); }
RPT_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

RPT_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* One Horner step p*t + c with a literal coefficient.  A 64-bit constant cannot be an inline operand; left to itself
 * the compiler materialises it with two v_mov into the destination of a two-address v_fmac_f64 — in the sky stage
 * those moves outnumbered the f64 arithmetic.  Asked for in an SGPR pair it costs two s_mov (scalar unit, off the
 * VALU port) and the step is one v_fma_f64.  Same operation, same bits. */
RPT_HD double horner(double p, double t, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "s"(c));
    return r;
#else
    return __builtin_fma(p, t, c);
#endif
}

/* ---- IEEE float helpers with Rust semantics ------------------------------ */
/* f32::min / f32::max: a NaN operand yields the other operand
 * (reference use: kernels/src/intersection.rs:107-116, bsdf.rs:275,303).
 * The sign of a zero result is never observable on the hot path (only compared
 * or re-clamped), so the device maps to v_min_f32 / v_max_f32. */
RPT_HD float fminr(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(a, b);
#else
    return (a < b || b != b) ? a : b;
#endif
}
RPT_HD float fmaxr(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(a, b);
#else
    return (a > b || b != b) ? a : b;
#endif
}
RPT_HD float sqrtr(float x) { return __builtin_sqrtf(x); }          /* IEEE correctly rounded on both sides */
/* (Measured and dropped, both exact where they claim to be — checked over all 2^32 floats on the device — and neither
 * faster: the compiler's correctly rounded expansion without its denormal pre-/post-scaling (11 instead of 16 instructions,
 * for arguments >= 2^-96: sky stage 49.2 -> 49.0 ms), and s = fma(x - s0^2, 0.5 * v_rcp_f32(s0), s0) behind a range test
 * (7 instructions; wrong on 100 floats; its branch made the sky stage slower).  The root is not what these stages wait for.) */
/* a / c for a compile-time constant c (rc = RN(1 / c)) and a numerator that is zero, not finite, or at least 2^-100 in
 * magnitude: Markstein's q0 = RN(a rc), r = a - q0 c (exact), q = RN(q0 + r rc) is the correctly rounded quotient, and
 * v_div_fixup_f32 puts back what the three steps lose on zeros and infinities (the sign of a zero, inf instead of NaN) —
 * 4 instructions without a branch instead of the 10 of an IEEE division.  Checked on the device for every float of that
 * domain per constant used (tests/test_gpu_math_exhaustive.py).  A TINY non-zero numerator is outside the contract (the
 * residual would underflow); callers state why theirs cannot be one.  (With a range test and an IEEE fallback instead of the
 * precondition the sky stage got SLOWER, 52.3 -> 55.3 ms: the branch splits the scheduling region of a latency-bound march.) */
RPT_HD float div_const_nontiny(float a, float c, float rc) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RPT_NO_DIV_CONST)
    const float q0 = a * rc;
    const float r = __builtin_fmaf(-q0, c, a);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(r, rc, q0), c, a);
#else
    (void)rc;
    return a / c;
#endif
}
/* x / 255.0f for x = 0.0 .. 255.0 (an atlas texel channel, src/asset.rs:270 `Vec4(r, g, b, 255) / 255.0`): the Markstein sequence above without its
 * fix-up — the numerator is zero or a small integer, never tiny, infinite or negative zero.  Correctly rounded on all 256 values
 * (all 256, host build and device: tests/test_math.py, tests/test_gpu_parity.py), in three instructions instead of the ten of an
 * IEEE division: a textured hit converts up to 48 channels (16 texels of four bilinear lookups). */
RPT_HD float unorm8(float x) {
    const float rc = 1.0f / 255.0f;
    const float q0 = x * rc;
    return __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, x), rc, q0);
}
RPT_HD float absr(float x) { return u2f(f2u(x) & 0x7fffffffu); }
RPT_HD bool finiter(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }
RPT_HD bool isnanr(float x) { return x != x; }
RPT_HD bool isinfr(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }
RPT_HD float floorr(float x) { return __builtin_floorf(x); }
RPT_HD float ceilr(float x) { return __builtin_ceilf(x); }

/* Rust `f32 as u32` / `f32 as usize`: saturating, NaN -> 0
 * (reference use: kernels/src/light_pick.rs:10, src/trace.rs:157). */
RPT_HD uint32_t f2u32_sat(float x) {
    if (!(x > 0.0f)) return 0u;             /* NaN, negatives, zero */
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}
/* Rust `f32 as i32` saturating, NaN -> 0 (image_polyfill.rs:41-42 as_ivec2) */
RPT_HD int32_t f2i32_sat(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    /* v_cvt_i32_f32 IS Rust's `as i32`: truncation, saturation at both ends, NaN -> 0 — one instruction instead of three compares and three selects around
     * it (a textured hit converts 16 footprint coordinates).  Equal to the branches below for every one of the 2^32 floats: rpt_debug_math_sweep op 2,
     * tests/test_gpu_math_exhaustive.py. */
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)x;
#endif
}
/* the branch form on both sides (what the instruction is checked against) */
RPT_HD int32_t f2i32_sat_reference(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)x;
}

/* f32::powi for the two constant exponents on the hot path. LLVM/compiler-rt
 * square-and-multiply: powi(x,2) = x*x ; powi(x,5) = x * ((x*x)*(x*x))
 * (reference use: kernels/src/util.rs:230,234-235, light_pick.rs:78). */
RPT_HD float powi2(float x) { return x * x; }
RPT_HD float powi5(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }

/* ---- double building blocks (add/mul/fma only) --------------------------- */

/* round-to-nearest-even integer of t, |t| < 2^51, via the 1.5*2^52 trick */
RPT_HD double rint_magic(double t) {
    const double M = 6755399441055744.0;
    double s = t + M;            /* not foldable without -ffast-math (never used here) */
    return s - M;
}

/* 2^k as a double, k in [-1022, 1023] */
RPT_HD double exp2i(int k) { return u2d((uint64_t)(k + 1023) << 52); }

/* reciprocal of d (normal, float-range magnitude) to ~1e-16 relative:
 * float seed (IEEE f32 divide) + two fma Newton steps. */
RPT_HD double rcp_newton(double d) {
    double y = (double)(1.0f / (float)d);
    double e = fmad(-d, y, 1.0);
    y = fmad(y, e, y);
    e = fmad(-d, y, 1.0);
    y = fmad(y, e, y);
    return y;
}

/* sqrt of z (z > 0, float-range) to ~1e-16 relative: float rsqrt seed
 * (two IEEE f32 ops) + two Newton steps on 1/sqrt + one Heron-style fix. */
RPT_HD double sqrt_newton(double z) {
    float zf = (float)z;
    double y = (double)(1.0f / __builtin_sqrtf(zf));
    double e = fmad(-(z * y), y, 1.0);
    y = fmad(y * 0.5, e, y);
    e = fmad(-(z * y), y, 1.0);
    y = fmad(y * 0.5, e, y);
    double s = z * y;
    double r = fmad(-s, s, z);
    s = fmad(r, y * 0.5, s);
    return s;
}

/* reduce x to r in [-pi/4, pi/4] with quadrant q (x ~= q*pi/2 + r) */
RPT_HD double reduce_pio2(double x, int &q) {
    double n = rint_magic(x * RPT_TWO_OVER_PI);
    double r = fmad(-n, RPT_PIO2_1, x);
    r = fmad(-n, RPT_PIO2_2, r);
    r = fmad(-n, RPT_PIO2_3, r);
    q = (int)((long long)n & 3);
    return r;
}

RPT_HD double sin_poly(double r) {
    RPT_COEF(k_sin_c, RPT_SIN_C);
    double t = r * r;
    double p = c[RPT_SIN_C_N - 1];
RPT_UNROLL
    for (int i = RPT_SIN_C_N - 2; i >= 1; --i) p = horner(p, t, c[i]);
    /* sin r = r + r*t*p */
    return fmad(r * t, p, r);
}
RPT_HD double cos_poly(double r) {
    RPT_COEF(k_cos_c, RPT_COS_C);
    double t = r * r;
    double p = c[RPT_COS_C_N - 1];
RPT_UNROLL
    for (int i = RPT_COS_C_N - 2; i >= 0; --i) p = horner(p, t, c[i]);
    return p;
}

/* sin and cos of a float angle; correctly rounded for |x| < 1e5 (the hot path
 * passes [0, 2pi], [0, pi/2] and the user's camera angles).  |x| >= 2^30 is
 * outside the supported domain and yields (0, 1) deterministically. */
RPT_HD void sincosr(float xf, float &s, float &c) {
    if (!finiter(xf)) { s = c = u2f(0x7fc00000u); return; }
    if (absr(xf) >= 1073741824.0f) { s = 0.0f; c = 1.0f; return; }
    if (xf == 0.0f) { s = xf; c = 1.0f; return; }           /* keeps sin(-0) = -0 */
    int q;
    double r = reduce_pio2((double)xf, q);
    double sr = sin_poly(r), cr = cos_poly(r);
    /* The quadrant fix-up on the ROUNDED floats, as integer logic: rounding to nearest commutes with negation and with picking one of the
     * two values, so the result is the same float as selecting / negating the doubles first — and the four back-to-back VOP2 v_cndmask (vcc)
     * a select of two doubles compiles to issue at ~30 cycles each on gfx950 (tools/microbench/valu_rates.hip: "cmp; 32 cnd vcc"), xor / and
     * at 2.  swap: (a, b) -> (a ^ d, b ^ d) with d = (a ^ b) & mask. */
    const uint32_t sb = f2u((float)sr), cb = f2u((float)cr);
    const uint32_t d = (sb ^ cb) & (0u - ((uint32_t)q & 1u));
    s = u2f((sb ^ d) ^ (((uint32_t)q & 2u) << 30));
    c = u2f((cb ^ d) ^ ((((uint32_t)q + 1u) & 2u) << 30));
}
RPT_HD float sinr(float x) { float s, c; sincosr(x, s, c); return s; }
RPT_HD float cosr(float x) { float s, c; sincosr(x, s, c); return c; }

/* exp of a double t, returned as double (t in [-700, 700]) */
RPT_HD double exp_core(double t) {
    RPT_COEF(k_exp_c, RPT_EXP_C);
    double kf = rint_magic(t * RPT_LOG2E);
    double r = fmad(-kf, RPT_LN2_HI, t);
    r = fmad(-kf, RPT_LN2_LO, r);
    double p = c[RPT_EXP_C_N - 1];
RPT_UNROLL
    for (int i = RPT_EXP_C_N - 2; i >= 0; --i) p = horner(p, r, c[i]);
    int k = (int)kf;
    /* split the scale so that 2^k never leaves the normal double range */
    int k1 = k / 2, k2 = k - k1;
    return (p * exp2i(k1)) * exp2i(k2);
}

RPT_HD float expr(float x) {
    if (x != x) return x;
    if (x > 89.0f) return u2f(0x7f800000u);
    if (x < -104.0f) return 0.0f;
    return (float)exp_core((double)x);
}

/* exp for the SKY march (reference use: kernels/src/skybox.rs:37-38, 62 — 84 calls per miss, 30 % of PBRTest's time
 * when they went through exp_core): float arithmetic only, 14 instructions instead of ~35 f64-heavy ones.  Deterministic
 * like everything here (IEEE mul / add / fma, explicit fma only), within 1 ulp of the exact value (tests/test_math.py:
 * max 1 ulp from the correctly rounded expr, equal on ~90 % of arguments) — NOT correctly rounded, and it need not be:
 * the sky's radiance ends a path, no later decision of that path (lobe, light pick, roulette, hit / miss) depends on
 * it, so a last-bit difference stays a 1e-7 relative difference of a pixel's value instead of a flipped path.
 * RPT_SKY_EXACT_EXP = 1 routes it back through the correctly rounded expr. */
#ifndef RPT_SKY_EXACT_EXP
#define RPT_SKY_EXACT_EXP 0
#endif
RPT_HD float exp2i_f(int k) { return u2f((uint32_t)(k + 127) << 23); }       /* 2^k, k in [-126, 127] */
RPT_HD float exp_sky(float x) {
#if RPT_SKY_EXACT_EXP
    return expr(x);
#else
    const float c[RPT_EXPF_C_N] = RPT_EXPF_C_INIT;
    /* Specials without their own branches (84 calls per sky miss: three compares, three selects and their constants were a third of this function): the
     * argument is clamped to [-104, 89] — at 89 the two scaling multiplies below overflow to +inf (1.32 * 2^128), at -104 they round 0.97 * 2^-150 to +0,
     * exactly what x > 89 and x < -104 must return — and a NaN is passed through by ONE select at the end.  Checked against the branching form by
     * tools/exp_sky_check.cpp: every one of the 2^32 floats in a manual run (stride 1, host build), stride 64 plus the windows around every boundary on each
     * test run (tests/test_math.py::test_exp_sky_specials_without_branches); the device at the special arguments in tests/test_gpu_math_exhaustive.py. */
    const float x_in = x;
    x = fmaxr(fminr(x, 89.0f), -104.0f);
    const float M = 12582912.0f;                        /* 1.5 * 2^23: round to nearest even integer, |t| < 2^22 */
    float kf = (x * RPT_LOG2E_F + M) - M;
    float r = __builtin_fmaf(-kf, RPT_LN2_HI_F, x);
    r = __builtin_fmaf(-kf, RPT_LN2_LO_F, r);
    float p = c[RPT_EXPF_C_N - 1];
RPT_UNROLL
    for (int i = RPT_EXPF_C_N - 2; i >= 0; --i) p = __builtin_fmaf(p, r, c[i]);
    p = __builtin_fmaf(r * r, p, r) + 1.0f;             /* 1 + r + r^2 P(r) */
    /* p * 2^k with ONE rounding, into the denormals and over the top alike: ldexp — a single v_ldexp_f32 on the device.  Round 5: it replaces 2^k in two
     * normal factors built with integer shifts and two multiplies ((p * 2^(k/2)) * 2^(k - k/2): the first product exact, the second rounding once — the
     * same value for every float, tools/exp_sky_check.cpp compares the two forms), ten instructions of this function's twenty-four, 84 calls per sky miss. */
    const float e = __builtin_ldexpf(p, (int)kf);
    return x_in != x_in ? x_in : e;
#endif
}

/* natural log of a positive finite double-representable float value */
RPT_HD double log_core(double x) {
    RPT_COEF(k_atanh_c, RPT_ATANH_C);
    uint64_t u = d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);   /* [1,2) */
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double f = (m - 1.0) * rcp_newton(m + 1.0);
    double t = f * f;
    double p = c[RPT_ATANH_C_N - 1];
RPT_UNROLL
    for (int i = RPT_ATANH_C_N - 2; i >= 1; --i) p = horner(p, t, c[i]);
    /* atanh f = f + f*t*p ; log m = 2 atanh f */
    double lm = 2.0 * fmad(f * t, p, f);
    double ed = (double)e;
    return fmad(ed, RPT_LN2_HI, fmad(ed, RPT_LN2_LO, lm));
}

/* f32::powf (reference use: kernels/src/skybox.rs:90,93). Full IEEE special
 * cases for the shapes that can occur: any x, finite y. */
RPT_HD float powr(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    bool y_is_int = (floorr(y) == y) && finiter(y);
    bool y_is_odd = y_is_int && (absr(y) < 16777216.0f) && (((long long)y) & 1);
    if (isinfr(y)) {
        float ax = absr(x);
        if (ax == 1.0f) return 1.0f;
        return ((ax > 1.0f) == (y > 0.0f)) ? u2f(0x7f800000u) : 0.0f;
    }
    if (x == 0.0f) {
        float z = (y > 0.0f) ? 0.0f : u2f(0x7f800000u);
        return (y_is_odd && (f2u(x) >> 31)) ? -z : z;
    }
    if (isinfr(x)) {
        float z = (y > 0.0f) ? u2f(0x7f800000u) : 0.0f;
        return (x < 0.0f && y_is_odd) ? -z : z;
    }
    float sign = 1.0f;
    if (x < 0.0f) {
        if (!y_is_int) return u2f(0x7fc00000u);
        if (y_is_odd) sign = -1.0f;
        x = -x;
    }
    double t = (double)y * log_core((double)x);
    if (t > 89.0) return sign * u2f(0x7f800000u);
    if (t < -104.0) return sign * 0.0f;
    return sign * (float)exp_core(t);
}

/* asin core: x * P(x^2), |x| <= 1/2 */
RPT_HD double asin_poly(double x, double t) {
    RPT_COEF(k_asin_c, RPT_ASIN_C);
    double p = c[RPT_ASIN_C_N - 1];
RPT_UNROLL
    for (int i = RPT_ASIN_C_N - 2; i >= 1; --i) p = horner(p, t, c[i]);
    return fmad(x * t, p, x * c[0]);
}

/* f32::acos (reference use: kernels/src/util.rs:25) */
RPT_HD float acosr(float xf) {
    if (xf != xf) return xf;
    float ax = absr(xf);
    if (ax > 1.0f) return u2f(0x7fc00000u);
    double x = (double)xf;
    if (ax <= 0.5f) {
        double a = asin_poly(x, x * x);
        return (float)((RPT_PIO2_D - a) + RPT_PIO2_LO);
    }
    double z = (1.0 - (double)ax) * 0.5;       /* exact */
    double two_asin = 0.0;
    if (z > 0.0) {
        double s = sqrt_newton(z);
        two_asin = 2.0 * asin_poly(s, z);
    }
    if (xf > 0.0f) return (float)two_asin;
    return (float)((RPT_PI_D - two_asin) + RPT_PI_LO);
}

/* f32::asin (reference use: kernels/src/lib.rs:75, image-skybox branch) */
RPT_HD float asinr(float xf) {
    if (xf != xf) return xf;
    float ax = absr(xf);
    if (ax > 1.0f) return u2f(0x7fc00000u);
    if (ax <= 0.5f) {
        if (xf == 0.0f) return xf;
        double x = (double)xf;
        return (float)asin_poly(x, x * x);
    }
    double z = (1.0 - (double)ax) * 0.5;
    double two_asin = 0.0;
    if (z > 0.0) {
        double s = sqrt_newton(z);
        two_asin = 2.0 * asin_poly(s, z);
    }
    double r = (RPT_PIO2_D - two_asin) + RPT_PIO2_LO;
    return (float)((xf < 0.0f) ? -r : r);
}

/* atan of a in [0, 1] as double */
RPT_HD double atan01(double a) {
    RPT_COEF(k_atan_c, RPT_ATAN_C);
    double base = 0.0, base_lo = 0.0, x = a;
    if (a > RPT_TAN_PIO8) {
        x = (a - 1.0) * rcp_newton(a + 1.0);
        base = RPT_PIO4_D;
        base_lo = RPT_PIO4_LO;
    }
    double t = x * x;
    double p = c[RPT_ATAN_C_N - 1];
RPT_UNROLL
    for (int i = RPT_ATAN_C_N - 2; i >= 1; --i) p = horner(p, t, c[i]);
    double at = fmad(x * t, p, x * c[0]);
    return (base + at) + base_lo;
}

/* f32::atan2 (reference use: kernels/src/lib.rs:72,74, image-skybox branch) */
RPT_HD float atan2r(float yf, float xf) {
    if (xf != xf || yf != yf) return u2f(0x7fc00000u);
    bool xneg = (f2u(xf) >> 31) != 0, yneg = (f2u(yf) >> 31) != 0;
    float ax = absr(xf), ay = absr(yf);
    double r;
    if (ay == 0.0f) {
        r = xneg ? RPT_PI_D : 0.0;
    } else if (ax == 0.0f) {
        r = RPT_PIO2_D;
    } else if (isinfr(ax) || isinfr(ay)) {
        if (isinfr(ax) && isinfr(ay)) r = xneg ? 3.0 * RPT_PIO4_D : RPT_PIO4_D;
        else if (isinfr(ay)) r = RPT_PIO2_D;
        else r = xneg ? RPT_PI_D : 0.0;
    } else {
        /* scale both into the normal float range so the float-seeded
         * reciprocal is well defined (ratio unchanged: powers of two) */
        double dx = (double)ax, dy = (double)ay;
        double hi = dx > dy ? dx : dy, lo = dx > dy ? dy : dx;
        uint64_t hb = d2u(hi) & 0x7ff0000000000000ull;
        double scale = u2d(0x7fe0000000000000ull - hb);      /* 2^(1023-e) * 2^-... keeps hi in [1,2) */
        hi *= scale; lo *= scale;
        double a = lo * rcp_newton(hi);
        double at = atan01(a);
        if (dy > dx) at = (RPT_PIO2_D - at) + RPT_PIO2_LO;
        if (xneg) at = (RPT_PI_D - at) + RPT_PI_LO;
        r = at;
    }
    float rf = (float)r;
    return yneg ? -rf : rf;
}

/* f32::atan (reference use: kernels/src/util.rs:125, Glass only): atan2(x, 1) is the same real number, and both are
 * rounded once from the same double evaluation */
RPT_HD float atanr(float x) { return atan2r(x, 1.0f); }

} /* namespace rptm */
#endif /* RPT_MATH_H */
