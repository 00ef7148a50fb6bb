// Issue cost of VALU instructions on gfx950, in SIMD cycles per wave64 instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
// Each kernel runs ITER trips of a 32-instruction unrolled body on 8 independent destination registers;
// grid = 256 CUs x 4 SIMDs x WPS waves.  cycles = time * f_clk * n_simd * / wave-instructions; the clock is
// calibrated on v_fma_f32 (assumed 4.0 cycles: 16 lanes per SIMD cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 16384

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEFINE_KERNEL(NAME, ASM_LINE, CLOBBER)                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out) {                                               \
        float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f;                                        \
        for (int i = 0; i < ITER; ++i) {                                                                    \
            asm volatile(ASM_LINE : "+v"(a) : "v"(b), "v"(c) : CLOBBER);                                    \
        }                                                                                                   \
        if (a == 12345.678f) out[0] = a;                                                                    \
    }

// 32 instructions per asm block, destination registers v[40..47] (+pairs up to v[40..55]) clobbered
#define R8(fmt_pre, fmt_post) \
    fmt_pre "40" fmt_post "\n" fmt_pre "41" fmt_post "\n" fmt_pre "42" fmt_post "\n" fmt_pre "43" fmt_post "\n" \
    fmt_pre "44" fmt_post "\n" fmt_pre "45" fmt_post "\n" fmt_pre "46" fmt_post "\n" fmt_pre "47" fmt_post "\n"
#define R32(p, q) R8(p, q) R8(p, q) R8(p, q) R8(p, q)
#define P8(fmt_pre, fmt_post) \
    fmt_pre "[40:41]" fmt_post "\n" fmt_pre "[42:43]" fmt_post "\n" fmt_pre "[44:45]" fmt_post "\n" fmt_pre "[46:47]" fmt_post "\n" \
    fmt_pre "[48:49]" fmt_post "\n" fmt_pre "[50:51]" fmt_post "\n" fmt_pre "[52:53]" fmt_post "\n" fmt_pre "[54:55]" fmt_post "\n"
#define P32(p, q) P8(p, q) P8(p, q) P8(p, q) P8(p, q)
#define CL "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s40","s41","s42","s43"
#define CLS "scc", CL      /* kernels with SALU instructions: they write SCC, which the loop's own compare may keep live across the asm block */

DEFINE_KERNEL(k_fma_f32, R32("v_fma_f32 v", ", %0, %1, %2"), CL)
DEFINE_KERNEL(k_mul_f32, R32("v_mul_f32 v", ", %0, %1"), CL)
DEFINE_KERNEL(k_add_f32, R32("v_add_f32 v", ", %0, %1"), CL)
DEFINE_KERNEL(k_min_f32, R32("v_min_f32 v", ", %0, %1"), CL)
DEFINE_KERNEL(k_max3_f32, R32("v_max3_f32 v", ", %0, %1, %2"), CL)
DEFINE_KERNEL(k_mov_b32, R32("v_mov_b32 v", ", %0"), CL)
DEFINE_KERNEL(k_cndmask, R32("v_cndmask_b32 v", ", %0, %1, vcc"), CL)
DEFINE_KERNEL(k_cndmask_sgpr, R32("v_cndmask_b32 v", ", %0, %1, s[42:43]"), CL)
DEFINE_KERNEL(k_cmp_vcc, R32("v_cmp_lt_f32 vcc, %0, v", ""), CL)
DEFINE_KERNEL(k_cmp_sgpr, R32("v_cmp_lt_f32 s[40:41], %0, v", ""), CL)
DEFINE_KERNEL(k_and_b32, R32("v_and_b32 v", ", %0, %1"), CL)
DEFINE_KERNEL(k_lshl_add, R32("v_lshl_add_u32 v", ", %0, 3, %1"), CL)
DEFINE_KERNEL(k_bfe, R32("v_bfe_u32 v", ", %0, 3, 8"), CL)
DEFINE_KERNEL(k_mul_lo_u32, R32("v_mul_lo_u32 v", ", %0, %1"), CL)
DEFINE_KERNEL(k_rcp_f32, R32("v_rcp_f32 v", ", %0"), CL)
DEFINE_KERNEL(k_sqrt_f32, R32("v_sqrt_f32 v", ", %0"), CL)
DEFINE_KERNEL(k_div_scale, R32("v_div_scale_f32 v", ", vcc, %0, %1, %0"), CL)
DEFINE_KERNEL(k_div_fmas, R32("v_div_fmas_f32 v", ", %0, %1, %2"), CL)
DEFINE_KERNEL(k_div_fixup, R32("v_div_fixup_f32 v", ", %0, %1, %2"), CL)
DEFINE_KERNEL(k_cvt_f64_f32, P32("v_cvt_f64_f32 v", ", %0"), CL)
DEFINE_KERNEL(k_pk_fma_f32, P32("v_pk_fma_f32 v", ", v[56:57], v[58:59], v[56:57]"), CL)
DEFINE_KERNEL(k_pk_mul_f32, P32("v_pk_mul_f32 v", ", v[56:57], v[58:59]"), CL)
DEFINE_KERNEL(k_pk_add_f32, P32("v_pk_add_f32 v", ", v[56:57], v[58:59]"), CL)
DEFINE_KERNEL(k_pk_mov_b32, P32("v_pk_mov_b32 v", ", v[56:57], v[58:59]"), CL)
DEFINE_KERNEL(k_fma_f64, P32("v_fma_f64 v", ", v[56:57], v[58:59], v[56:57]"), CL)
DEFINE_KERNEL(k_mul_f64, P32("v_mul_f64 v", ", v[56:57], v[58:59]"), CL)
DEFINE_KERNEL(k_add_f64, P32("v_add_f64 v", ", v[56:57], v[58:59]"), CL)
DEFINE_KERNEL(k_rcp_f64, P32("v_rcp_f64 v", ", v[56:57]"), CL)
DEFINE_KERNEL(k_cvt_f32_f64, R32("v_cvt_f32_f64 v", ", v[56:57]"), CL)
DEFINE_KERNEL(k_mad_u64_u32, P32("v_mad_u64_u32 v", ", vcc, %0, %1, v[56:57]"), CL)
DEFINE_KERNEL(k_lshl_add_u64, P32("v_lshl_add_u64 v", ", v[56:57], 3, v[58:59]"), CL)
DEFINE_KERNEL(k_ds_bpermute, R32("ds_bpermute_b32 v", ", %0, %1"), CL)
DEFINE_KERNEL(k_readlane, R32("v_readlane_b32 s42, %0, 3 ; v", ""), CL)
DEFINE_KERNEL(k_mbcnt, R32("v_mbcnt_lo_u32_b32 v", ", -1, 0"), CL)
DEFINE_KERNEL(k_s_and, R32("s_and_b64 s[40:41], s[42:43], exec ; v", ""), CLS)


#define PAIR8(a, b) a "40" b "40, %0, %1, " "\n" a "41" b "41, %0, %1, " "\n"
DEFINE_KERNEL(k_cmp_cnd_vcc, R8("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 v", ", %0, %1, vcc") R8("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 v", ", %0, %1, vcc"), CL)
DEFINE_KERNEL(k_cmp_cnd_sgpr, R8("v_cmp_lt_f32 s[40:41], %0, %1\n v_cndmask_b32 v", ", %0, %1, s[40:41]") R8("v_cmp_lt_f32 s[40:41], %0, %1\n v_cndmask_b32 v", ", %0, %1, s[40:41]"), CL)
DEFINE_KERNEL(k_cnd_e64_vcc, R32("v_cndmask_b32_e64 v", ", %0, %1, vcc"), CL)
DEFINE_KERNEL(k_cnd_after_cmp, "v_cmp_lt_f32 vcc, %0, %1\n" R32("v_cndmask_b32 v", ", %0, %1, vcc"), CL)
DEFINE_KERNEL(k_cnd_exec_full, "s_mov_b64 vcc, exec\n" R32("v_cndmask_b32 v", ", %0, %1, vcc"), CL)
DEFINE_KERNEL(k_cnd_zero, "s_mov_b64 vcc, 0\n" R32("v_cndmask_b32 v", ", %0, %1, vcc"), CL)
DEFINE_KERNEL(k_min_fma_mix, R8("v_min_f32 v", ", %0, %1\n v_fma_f32 v48, %0, %1, %2\n v_fma_f32 v49, %0, %1, %2\n v_fma_f32 v50, %0, %1, %2") , CL)
DEFINE_KERNEL(k_fma_x32_dep, R32("v_fma_f32 %0, %0, %1, %2 ; v", ""), CL)


#define CMPV "v_cmp_lt_f32 vcc, %0, %1\n"
#define CND(n) "v_cndmask_b32 v" #n ", %0, %1, vcc\n"
#define FMA(n) "v_fma_f32 v" #n ", %0, %1, %2\n"
#define MIN(n) "v_min_f32 v" #n ", %0, %1\n"
// 8 x (cmp + 3 cnd) = 32 instr
DEFINE_KERNEL(k_cmp_3cnd, CMPV CND(40) CND(41) CND(42) CMPV CND(43) CND(44) CND(45) CMPV CND(46) CND(47) CND(40) CMPV CND(41) CND(42) CND(43) CMPV CND(44) CND(45) CND(46) CMPV CND(47) CND(40) CND(41) CMPV CND(42) CND(43) CND(44) CMPV CND(45) CND(46) CND(47), CL)
// 10 x (cmp + 2 cnd) + 2 fma = 32
DEFINE_KERNEL(k_cmp_2cnd, CMPV CND(40) CND(41) CMPV CND(42) CND(43) CMPV CND(44) CND(45) CMPV CND(46) CND(47) CMPV CND(40) CND(41) CMPV CND(42) CND(43) CMPV CND(44) CND(45) CMPV CND(46) CND(47) CMPV CND(40) CND(41) CMPV CND(42) CND(43) FMA(44) FMA(45), CL)
// 16 x (cnd, fma)
DEFINE_KERNEL(k_cnd_fma, CND(40) FMA(48) CND(41) FMA(49) CND(42) FMA(50) CND(43) FMA(51) CND(44) FMA(48) CND(45) FMA(49) CND(46) FMA(50) CND(47) FMA(51) CND(40) FMA(48) CND(41) FMA(49) CND(42) FMA(50) CND(43) FMA(51) CND(44) FMA(48) CND(45) FMA(49) CND(46) FMA(50) CND(47) FMA(51), CL)
// 16 x (cnd, min)
DEFINE_KERNEL(k_cnd_min, CND(40) MIN(48) CND(41) MIN(49) CND(42) MIN(50) CND(43) MIN(51) CND(44) MIN(48) CND(45) MIN(49) CND(46) MIN(50) CND(47) MIN(51) CND(40) MIN(48) CND(41) MIN(49) CND(42) MIN(50) CND(43) MIN(51) CND(44) MIN(48) CND(45) MIN(49) CND(46) MIN(50) CND(47) MIN(51), CL)


// ---- item 7 of round 1's review: would an f32-only (double-float) build of the shared transcendentals be cheaper than the
// f64 Horner chains rpt_math.h uses?  One Horner step p = p * t + c, 8 independent chains x 4 steps per trip, (a) as one
// v_fma_f64, (b) in double-float arithmetic on float pairs (two_prod by fma, two_sum, renormalise: 16 f32 instructions).
__global__ __launch_bounds__(256) void k_horner_f64(float *out) {
    double p[8], t = 0.999999 + threadIdx.x * 1e-9, c = 1e-3;
    for (int k = 0; k < 8; ++k) p[k] = 1.0 + k * 1e-3 + threadIdx.x * 1e-6;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = __builtin_fma(p[k], t, c);
    }
    double acc = 0.0;
    for (int k = 0; k < 8; ++k) acc += p[k];
    if (acc == 12345.678) out[0] = (float)acc;
}
__global__ __launch_bounds__(256) void k_horner_df32(float *out) {
    float ph[8], pl[8];
    const float th = 0.999999f + threadIdx.x * 1e-9f, tl = 1e-9f, ch = 1e-3f, cl = 1e-11f;
    for (int k = 0; k < 8; ++k) { ph[k] = 1.0f + k * 1e-3f + threadIdx.x * 1e-6f; pl[k] = 1e-9f; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float hh = ph[k] * th, e = __builtin_fmaf(ph[k], th, -hh);           /* two_prod */
                float lo = e + __builtin_fmaf(ph[k], tl, pl[k] * th);
                float s = hh + ch, bb = s - hh, err = (hh - (s - bb)) + (ch - bb);   /* two_sum */
                float l2 = err + (lo + cl);
                float rh = s + l2;
                ph[k] = rh;
                pl[k] = l2 - (rh - s);
            }
    }
    float acc = 0.0f;
    for (int k = 0; k < 8; ++k) acc += ph[k] + pl[k];
    if (acc == 12345.678f) out[0] = acc;
}

// ---- does a wave64 instruction get cheaper when whole 16- or 32-lane groups of EXEC are off?  (It would make "pack the live
// lanes of a wave into its lower half" worth building for the streamed traversal.)
#define DEFINE_MASKED(NAME, LO, HI, ASM_LINE)                                                                 \
    __global__ __launch_bounds__(256) void NAME(float *out) {                                               \
        float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f;                                        \
        asm volatile("s_mov_b32 exec_lo, " LO "\n s_mov_b32 exec_hi, " HI ::: "memory");                                                  \
        for (int i = 0; i < ITER; ++i) {                                                                    \
            asm volatile(ASM_LINE : "+v"(a) : "v"(b), "v"(c) : CL);                                         \
        }                                                                                                   \
        asm volatile("s_mov_b64 exec, -1" ::: "memory");                                                     \
        if (a == 12345.678f) out[0] = a;                                                                    \
    }
DEFINE_MASKED(k_fma_lo32, "-1", "0", R32("v_fma_f32 v", ", %0, %1, %2"))
DEFINE_MASKED(k_fma_lo16, "0xffff", "0", R32("v_fma_f32 v", ", %0, %1, %2"))
DEFINE_MASKED(k_fma_hi32, "0", "-1", R32("v_fma_f32 v", ", %0, %1, %2"))
DEFINE_MASKED(k_fma_alt, "0x55555555", "0x55555555", R32("v_fma_f32 v", ", %0, %1, %2"))
DEFINE_MASKED(k_fma_one, "1", "0", R32("v_fma_f32 v", ", %0, %1, %2"))
DEFINE_MASKED(k_min_lo32, "-1", "0", R32("v_min_f32 v", ", %0, %1"))
DEFINE_MASKED(k_min_lo16, "0xffff", "0", R32("v_min_f32 v", ", %0, %1"))
DEFINE_MASKED(k_min_alt, "0x55555555", "0x55555555", R32("v_min_f32 v", ", %0, %1"))
DEFINE_MASKED(k_f64_lo32, "-1", "0", P32("v_fma_f64 v", ", v[56:57], v[58:59], v[56:57]"))
DEFINE_MASKED(k_f64_lo16, "0xffff", "0", P32("v_fma_f64 v", ", v[56:57], v[58:59], v[56:57]"))
DEFINE_MASKED(k_rcp_lo32, "-1", "0", R32("v_rcp_f32 v", ", %0"))
DEFINE_MASKED(k_rcp_lo16, "0xffff", "0", R32("v_rcp_f32 v", ", %0"))


// ---- round 4: do the fma-class pipe (fma / mul / add / mov: ~2 cycles per wave64 instruction) and the "other" pipe (min / max3 / cmp /
// cndmask / integer: ~4) overlap ACROSS waves, or only between ADJACENT instructions of one wave?  The traversal's inner step is 48
// fma-class instructions in two runs of 24 followed by runs of max3 / min3 / cmp: if a run of one class blocks its pipe for every wave,
// re-ordering the body (sched_group_barrier) would pay.  Same 72 instructions, blocked (48 fma, then 24 min) vs interleaved (2 fma, 1 min).
#define F(n) "v_fma_f32 v" #n ", %0, %1, %2\n"
#define M(n) "v_min_f32 v" #n ", %0, %1\n"
#define F8 F(40) F(41) F(42) F(43) F(44) F(45) F(46) F(47)
#define M8 M(48) M(49) M(50) M(51) M(52) M(53) M(54) M(55)
DEFINE_KERNEL(k_blocked_48_24, F8 F8 F8 F8 F8 F8 M8 M8 M8, CL)
#define FFM(a, b, c) F(a) F(b) M(c)
DEFINE_KERNEL(k_inter_2_1, FFM(40, 41, 48) FFM(42, 43, 49) FFM(44, 45, 50) FFM(46, 47, 51) FFM(40, 41, 52) FFM(42, 43, 53) FFM(44, 45, 54) FFM(46, 47, 55)
                           FFM(40, 41, 48) FFM(42, 43, 49) FFM(44, 45, 50) FFM(46, 47, 51) FFM(40, 41, 52) FFM(42, 43, 53) FFM(44, 45, 54) FFM(46, 47, 55)
                           FFM(40, 41, 48) FFM(42, 43, 49) FFM(44, 45, 50) FFM(46, 47, 51) FFM(40, 41, 52) FFM(42, 43, 53) FFM(44, 45, 54) FFM(46, 47, 55), CL)
// the same with the dependences of a slab test: each min consumes the two fma results before it (interleaved), or all at the end (blocked)
#define MD(d, a, b) "v_min_f32 v" #d ", v" #a ", v" #b "\n"
DEFINE_KERNEL(k_blocked_dep, F8 F8 F8 F8 F8 F8 MD(48, 40, 41) MD(49, 42, 43) MD(50, 44, 45) MD(51, 46, 47) MD(52, 40, 41) MD(53, 42, 43) MD(54, 44, 45) MD(55, 46, 47)
                             MD(48, 40, 41) MD(49, 42, 43) MD(50, 44, 45) MD(51, 46, 47) MD(52, 40, 41) MD(53, 42, 43) MD(54, 44, 45) MD(55, 46, 47)
                             MD(48, 40, 41) MD(49, 42, 43) MD(50, 44, 45) MD(51, 46, 47) MD(52, 40, 41) MD(53, 42, 43) MD(54, 44, 45) MD(55, 46, 47), CL)
#define FFMD(a, b, c) F(a) F(b) MD(c, a, b)
DEFINE_KERNEL(k_inter_dep, FFMD(40, 41, 48) FFMD(42, 43, 49) FFMD(44, 45, 50) FFMD(46, 47, 51) FFMD(40, 41, 52) FFMD(42, 43, 53) FFMD(44, 45, 54) FFMD(46, 47, 55)
                           FFMD(40, 41, 48) FFMD(42, 43, 49) FFMD(44, 45, 50) FFMD(46, 47, 51) FFMD(40, 41, 52) FFMD(42, 43, 53) FFMD(44, 45, 54) FFMD(46, 47, 55)
                           FFMD(40, 41, 48) FFMD(42, 43, 49) FFMD(44, 45, 50) FFMD(46, 47, 51) FFMD(40, 41, 52) FFMD(42, 43, 53) FFMD(44, 45, 54) FFMD(46, 47, 55), CL)
// software-pipelined: the min of pair k issued after the fmas of pair k + 1 (what a scheduler would emit)
#define FFMP(a, b, c, pa, pb) F(a) F(b) MD(c, pa, pb)
DEFINE_KERNEL(k_inter_dep_lag, FFMP(40, 41, 48, 46, 47) FFMP(42, 43, 49, 40, 41) FFMP(44, 45, 50, 42, 43) FFMP(46, 47, 51, 44, 45) FFMP(40, 41, 52, 46, 47) FFMP(42, 43, 53, 40, 41) FFMP(44, 45, 54, 42, 43) FFMP(46, 47, 55, 44, 45)
                               FFMP(40, 41, 48, 46, 47) FFMP(42, 43, 49, 40, 41) FFMP(44, 45, 50, 42, 43) FFMP(46, 47, 51, 44, 45) FFMP(40, 41, 52, 46, 47) FFMP(42, 43, 53, 40, 41) FFMP(44, 45, 54, 42, 43) FFMP(46, 47, 55, 44, 45)
                               FFMP(40, 41, 48, 46, 47) FFMP(42, 43, 49, 40, 41) FFMP(44, 45, 50, 42, 43) FFMP(46, 47, 51, 44, 45) FFMP(40, 41, 52, 46, 47) FFMP(42, 43, 53, 40, 41) FFMP(44, 45, 54, 42, 43) FFMP(46, 47, 55, 44, 45), CL)


// ---- round 4: does the scalar unit take issue slots from the VALU?  The LDS walk issues 0.45 SALU per VALU instruction (SQ_INSTS_SALU / SQ_INSTS_VALU:
// mask logic, exec save / restore, loop control).  32 v_fma per trip alone, with 16 and with 32 independent s_and_b64 between them.

#define FS1(a) F(a) "s_and_b64 s[40:41], s[42:43], exec\n"
#define FFS(a, b) F(a) F(b) "s_and_b64 s[40:41], s[42:43], exec\n"
DEFINE_KERNEL(k_fma32_salu16, FFS(40, 41) FFS(42, 43) FFS(44, 45) FFS(46, 47) FFS(40, 41) FFS(42, 43) FFS(44, 45) FFS(46, 47)
                              FFS(40, 41) FFS(42, 43) FFS(44, 45) FFS(46, 47) FFS(40, 41) FFS(42, 43) FFS(44, 45) FFS(46, 47), CLS)
DEFINE_KERNEL(k_fma32_salu32, FS1(40) FS1(41) FS1(42) FS1(43) FS1(44) FS1(45) FS1(46) FS1(47) FS1(40) FS1(41) FS1(42) FS1(43) FS1(44) FS1(45) FS1(46) FS1(47)
                              FS1(40) FS1(41) FS1(42) FS1(43) FS1(44) FS1(45) FS1(46) FS1(47) FS1(40) FS1(41) FS1(42) FS1(43) FS1(44) FS1(45) FS1(46) FS1(47), CLS)
// and with the mask chained through the scalar unit into the VALU (v_cmp -> s_and -> v_cndmask with that mask), as the walk's predicates are
#define CSC(a) "v_cmp_lt_f32 s[40:41], %0, %1\n s_and_b64 s[42:43], s[40:41], exec\n v_cndmask_b32 v" #a ", %0, %1, s[42:43]\n"
DEFINE_KERNEL(k_cmp_sand_cnd, CSC(40) CSC(41) CSC(42) CSC(43) CSC(44) CSC(45) CSC(46) CSC(47) CSC(40) CSC(41) CSC(42) CSC(43) CSC(44) CSC(45) CSC(46) CSC(47), CLS)

struct Case { const char *name; void (*fn)(float *); int n_instr = 32; };

int main(int argc, char **argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 2;           // waves per SIMD
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wps;                     // 256 threads = 4 waves = one wave per SIMD per block
    float *out; hipMalloc(&out, 4);
    Case cases[] = {
        {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32}, {"v_min_f32", k_min_f32}, {"v_max3_f32", k_max3_f32},
        {"v_mov_b32", k_mov_b32}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32 sgpr", k_cndmask_sgpr}, {"v_cmp_lt_f32 vcc", k_cmp_vcc}, {"v_cmp_lt_f32 sgpr", k_cmp_sgpr},
        {"v_and_b32", k_and_b32}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfe_u32", k_bfe}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_div_scale_f32", k_div_scale}, {"v_div_fmas_f32", k_div_fmas},
        {"v_div_fixup_f32", k_div_fixup}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mov_b32", k_pk_mov_b32},
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"ds_bpermute_b32", k_ds_bpermute},
        {"v_readlane_b32", k_readlane}, {"16x(cmp vcc+cnd vcc)", k_cmp_cnd_vcc}, {"16x(cmp s+cnd s)", k_cmp_cnd_sgpr}, {"cnd e64 vcc", k_cnd_e64_vcc}, {"cmp; 32 cnd vcc", k_cnd_after_cmp}, {"vcc=exec; 32 cnd", k_cnd_exec_full}, {"vcc=0; 32 cnd", k_cnd_zero}, {"8x(min+3fma)", k_min_fma_mix}, {"fma dependent", k_fma_x32_dep}, {"8x(cmp+3cnd)", k_cmp_3cnd}, {"10x(cmp+2cnd)+2fma", k_cmp_2cnd}, {"16x(cnd,fma)", k_cnd_fma}, {"16x(cnd,min)", k_cnd_min}, {"v_mbcnt_lo", k_mbcnt}, {"s_and_b64", k_s_and},
        {"v_fma_f32 exec=lo32", k_fma_lo32}, {"v_fma_f32 exec=lo16", k_fma_lo16}, {"v_fma_f32 exec=hi32", k_fma_hi32}, {"v_fma_f32 exec=0x5555..", k_fma_alt}, {"v_fma_f32 exec=1", k_fma_one},
        {"v_min_f32 exec=lo32", k_min_lo32}, {"v_min_f32 exec=lo16", k_min_lo16}, {"v_min_f32 exec=0x5555..", k_min_alt},
        {"v_fma_f64 exec=lo32", k_f64_lo32}, {"v_fma_f64 exec=lo16", k_f64_lo16}, {"v_rcp_f32 exec=lo32", k_rcp_lo32}, {"v_rcp_f32 exec=lo16", k_rcp_lo16},
        {"72: 48 fma then 24 min (independent)", k_blocked_48_24, 72}, {"72: 24 x (2 fma, 1 min) (independent)", k_inter_2_1, 72},
        {"72: 48 fma then 24 min of their results", k_blocked_dep, 72}, {"72: 24 x (2 fma, min of the two)", k_inter_dep, 72}, {"72: 24 x (2 fma, min of the previous two)", k_inter_dep_lag, 72},
        {"32 fma + 16 s_and (cycles per VALU)", k_fma32_salu16}, {"32 fma + 32 s_and (cycles per VALU)", k_fma32_salu32}, {"16 x (cmp -> s_and -> cnd) (per VALU)", k_cmp_sand_cnd},
        {"Horner step f64 (1 v_fma_f64)", k_horner_f64}, {"Horner step double-float (16 f32)", k_horner_df32},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("waves/SIMD %d, %d CUs\n", wps, cus);
    auto time_of = [&](void (*fn)(float *)) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0); fn<<<blocks, 256>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        return best;
    };
    for (int w = 0; w < 20; ++w) k_fma_f32<<<blocks, 256>>>(out);     // clocks up
    hipDeviceSynchronize();
    for (auto &c : cases) {
        float base = time_of(k_fma_f32);
        float t = time_of(c.fn);
        float base2 = time_of(k_fma_f32);
        double rel = t / (0.5 * (base + base2)) * 4.0 * 32.0 / c.n_instr;
        printf("%-34s %8.3f ms (fma %.3f/%.3f)  %6.2f cycles/wave-instr (v_fma_f32 := 4)\n", c.name, t, base, base2, rel);
    }
    return 0;
}
