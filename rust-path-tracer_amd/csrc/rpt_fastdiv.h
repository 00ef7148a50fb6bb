/*
 * rpt_fastdiv.h — exact IEEE f32 division by a per-ray constant.
 *
 * intersect_aabb (reference: kernels/src/intersection.rs:104-122) performs six
 * TRUE divisions (bound - ro) / rd per box.  An IEEE f32 divide costs ~11 VALU
 * instructions on gfx950 (v_div_scale x2, v_rcp, 4-5 fma, v_div_fmas,
 * v_div_fixup) and the divisor only changes once per ray.  With ry = RN(1 / y)
 * (one real division per axis per ray), Markstein's sequence
 *     q0 = RN(x * ry);  r = x - q0 * y (exact, fma);  q = RN(q0 + r * ry)
 * returns RN(x / y) bit for bit as long as no intermediate leaves the normal
 * range.  The guards below guarantee that: |y| in [2^-40, 4); every ray-origin
 * component and every BVH bound is 0 or has magnitude in [2^-60, 2^40), so a
 * non-zero dividend b - o lies in [2^-84, 2^41].  Rays or scenes outside the
 * guards take the true-division path, so results never depend on which path ran.
 * (One representational difference: the dividend -0.0 over a positive divisor
 * gives +0.0 instead of -0.0.  The slab test only COMPARES these quotients, so
 * the sign of a zero is unobservable there.)
 * Evidence: tools/fastdiv_campaign.cpp (all 2^23 divisor mantissas x 254 dividend
 * patterns x 8 exponent/sign placements = 1.7e10 pairs, plus 2.4e9 random pairs:
 * 0 mismatches) and tests/test_fastdiv.py (CPU) / test_gpu_parity.py (device).
 */
#ifndef RPT_FASTDIV_H
#define RPT_FASTDIV_H

#include "rpt_math.h"

namespace rptm {

RPT_HD float div_by_rcp(float x, float y, float ry) {
    float q0 = x * ry;
    float r = __builtin_fmaf(-q0, y, x);
    return __builtin_fmaf(r, ry, q0);
}
/* divisor guard: finite, |y| in [2^-40, 4) */
RPT_HD bool fastdiv_divisor_ok(float y) {
    uint32_t e = (f2u(y) >> 23) & 0xffu;
    return e >= 127u - 40u && e <= 127u + 1u;
}
/* dividend-operand guard (ray origin / box bound): 0, or |v| in [2^-60, 2^40) */
RPT_HD bool fastdiv_operand_ok(float v) {
    uint32_t u = f2u(v) & 0x7fffffffu;
    uint32_t e = u >> 23;
    return u == 0u || (e >= 127u - 60u && e < 127u + 40u);
}
/* (b - o) / y exactly as the slab test computes it, choosing the fast path when allowed (test hook) */
RPT_HD float slab_quotient(float b, float o, float y) {
    float x = b - o;
    if (fastdiv_divisor_ok(y) && fastdiv_operand_ok(b) && fastdiv_operand_ok(o)) return div_by_rcp(x, y, 1.0f / y);
    return x / y;
}

}  // namespace rptm
#endif
