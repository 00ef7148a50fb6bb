// exp_sky of rpt_math.h (argument clamped to [-104, 89], one NaN select, 2^k applied with ldexp: round 5) == the form with explicit special-case branches and
// 2^k in two normal factors it replaced (rounds 4 / 5), for
// every float bit pattern (stride 1: 2^32 arguments, ~30 s on 8 threads) or every stride-th one plus all arguments within 2^20 patterns of the three
// boundaries.  usage: exp_sky_check [stride]   build: g++ -O2 -std=c++20 -ffp-contract=off -mfma -pthread -I<repo> -o exp_sky_check tools/exp_sky_check.cpp
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <cstdlib>
#include "rust-path-tracer_amd/csrc/rpt_math.h"
static inline float old_exp_sky(float x) {
    const float c[RPT_EXPF_C_N] = RPT_EXPF_C_INIT;
    if (x != x) return x;
    if (x > 89.0f) return rptm::u2f(0x7f800000u);
    if (x < -104.0f) return 0.0f;
    const float M = 12582912.0f;
    float kf = (x * RPT_LOG2E_F + M) - M;
    float r = __builtin_fmaf(-kf, RPT_LN2_HI_F, x);
    r = __builtin_fmaf(-kf, RPT_LN2_LO_F, r);
    float p = c[RPT_EXPF_C_N - 1];
    for (int i = RPT_EXPF_C_N - 2; i >= 0; --i) p = __builtin_fmaf(p, r, c[i]);
    p = __builtin_fmaf(r * r, p, r) + 1.0f;
    int k = (int)kf;
    int k1 = k / 2, k2 = k - k1;
    return (p * rptm::exp2i_f(k1)) * rptm::exp2i_f(k2);
}
int main(int argc, char **argv) {
    const uint64_t stride = argc > 1 ? (uint64_t)atoll(argv[1]) : 1;
    std::atomic<uint64_t> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t) th.emplace_back([&, t] {
        uint64_t b = 0;
        for (uint64_t u = (uint64_t)t << 29; u < ((uint64_t)(t + 1) << 29); ++u) {
            if (stride > 1 && u % stride != 0) {
                const uint32_t v = (uint32_t)u;     /* near 89.0f, -104.0f, +-inf / NaN: every pattern */
                const bool near = (v > 0x42b20000u - (1u << 20) && v < 0x42b20000u + (1u << 20)) || (v > 0xc2d00000u - (1u << 20) && v < 0xc2d00000u + (1u << 20)) ||
                                  (v & 0x7f800000u) == 0x7f800000u;
                if (!near) { u += stride - (u % stride) - 1; continue; }
            }
            float x = rptm::u2f((uint32_t)u);
            float a = old_exp_sky(x), n = rptm::exp_sky(x);
            if (rptm::f2u(a) != rptm::f2u(n)) { if (b < 5) printf("x=%08x old=%08x new=%08x\n", (uint32_t)u, rptm::f2u(a), rptm::f2u(n)); ++b; }
        }
        bad += b;
    });
    for (auto &x : th) x.join();
    printf("mismatches: %llu (stride %llu)\n", (unsigned long long)bad.load(), (unsigned long long)stride);
    return bad.load() != 0;
}
