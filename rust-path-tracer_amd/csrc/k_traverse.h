/*
 * k_traverse.h — BVH traversal + ray/triangle intersection stage.
 *
 * Produces, per ray, exactly what the reference's stack traversal produces
 * (kernels/src/intersection.rs:177-234, with intersect_aabb :104-122 and
 * muller_trumbore :9-54): same visiting order, same strict comparisons, same
 * f32 operations — so t / triangle index / backface are bit-identical to the
 * CPU path.  What differs is the machinery:
 *   - one wave64 lane per ray, rays read as two coalesced float4 records;
 *   - the traversal stack lives in LDS, laid out [entry][lane] so that every
 *     ds_read/ds_write of a wave is bank-conflict free regardless of the
 *     per-lane stack depth (lane l always hits bank l mod 32 of its half);
 *   - the near child is entered directly (the reference pushes it and pops it
 *     straight back, :229 then :183), only the far child is pushed;
 *   - both children of an inner node are adjacent (right = left + 1), so one
 *     node visit is 4 float4 loads = 64 contiguous bytes;
 *   - leaf triangles come from a pre-gathered (a, b-a, c-a) array instead of
 *     index buffer -> 3 x 64-byte vertex records (same f32 subtractions, done
 *     once at upload).
 * Traversal order is part of the NEAREST-hit result (ties in t keep the first
 * triangle visited): no reordering there.  The ANY-hit walk (light_pick.rs:141-148
 * + intersection.rs:173-234 with NEAREST_HIT = false) must match the reference in
 * `.hit` ONLY — light_pick.rs:148 reads nothing else: result.t stays 1e6 until the
 * first accept, which returns (:191-203), boxes are pruned against that constant
 * (:212-213: no "max_t" box pruning), so the set of boxes a ray may enter and with
 * it `.hit` are independent of the order siblings are visited in.  FIXED = true
 * walks a copy of the tree whose pairs were flipped at upload (shadow_order.h: the
 * preferred child in the left slot) left-first: no `tl > tr`, no swap.
 */
#ifndef RPT_K_TRAVERSE_H
#define RPT_K_TRAVERSE_H

#include "k_common.h"
#include "k_path.h"
#include "rpt_fastdiv.h"

template <bool SMALL> struct StackElem { typedef uint32_t type; };      /* node indices on the stack */
template <> struct StackElem<true> { typedef uint16_t type; };          /* < 65 536 nodes (and every LDS-resident scene) */
/* The stack of a lane is a column of an LDS array, [entry][lane].  Entry widths: 16 bits (< 65 536 nodes and every
 * LDS-resident scene), 32 bits, and — for the streamed global-memory walks, whose occupancy the LDS footprint caps once
 * indices need more than 16 bits — 24 bits as a 16-bit and an 8-bit column (< 2^24 nodes): 32 entries cost a wave 6 KB
 * instead of 8 KB, 5.5 instead of 4.25 waves per SIMD fit beside each other. */
template <typename T> __device__ __forceinline__ void stack_put(T *s, int sp, uint32_t v) { s[sp * RPT_WAVE] = (T)v; }
template <typename T> __device__ __forceinline__ uint32_t stack_get(T *s, int sp) { return (uint32_t)s[sp * RPT_WAVE]; }
struct Stack24 {
    uint16_t *lo;
    uint8_t *hi;
};
__device__ __forceinline__ void stack_put(Stack24 s, int sp, uint32_t v) {
    s.lo[sp * RPT_WAVE] = (uint16_t)v;
    s.hi[sp * RPT_WAVE] = (uint8_t)(v >> 16);
}
__device__ __forceinline__ uint32_t stack_get(Stack24 s, int sp) { return (uint32_t)s.lo[sp * RPT_WAVE] | ((uint32_t)s.hi[sp * RPT_WAVE] << 16); }
/* 16 + K bits per entry at the LDS cost of 16: the low half in the 16-bit column, bit 16 + k of level sp in bit sp of a per-lane
 * mask register (stacks have at most 32 levels).  One LDS operation per push / pop like the 16-bit stack, a few VALU
 * instructions per extra bit instead of the second column's LDS operation and its 1.5 KB per wave. */
template <int K> struct StackBits {
    uint16_t *lo;
    uint32_t hi[K];
};
template <int K> __device__ __forceinline__ void stack_put(StackBits<K> &s, int sp, uint32_t v) {
    s.lo[sp * RPT_WAVE] = (uint16_t)v;
    const uint32_t keep = ~(1u << sp);
#pragma unroll
    for (int k = 0; k < K; ++k) s.hi[k] = (s.hi[k] & keep) | (((v >> (16 + k)) & 1u) << sp);
}
template <int K> __device__ __forceinline__ uint32_t stack_get(const StackBits<K> &s, int sp) {
    uint32_t v = (uint32_t)s.lo[sp * RPT_WAVE];
#pragma unroll
    for (int k = 0; k < K; ++k) v |= ((s.hi[k] >> sp) & 1u) << (16 + k);
    return v;
}
/* the LDS arrays of one wave's stack for an entry width, and the handle walk_run takes */
template <int STACK, int WIDTH> struct WaveStack {                       /* WIDTH 16 / 32 */
    typedef typename StackElem<WIDTH == 16>::type T;
    T cells[STACK][RPT_WAVE];
    __device__ __forceinline__ T *column(uint32_t lane) { return &cells[0][lane]; }
};
template <int STACK> struct WaveStack<STACK, 21> {                      /* < 2^21 nodes: 16 bits in LDS + 5 mask registers */
    static_assert(STACK <= 32, "one mask bit per stack level");
    uint16_t lo[STACK][RPT_WAVE];
    __device__ __forceinline__ StackBits<5> column(uint32_t lane) { return StackBits<5>{&lo[0][lane], {0u, 0u, 0u, 0u, 0u}}; }
};
template <int STACK> struct WaveStack<STACK, 24> {
    uint16_t lo[STACK][RPT_WAVE];
    uint8_t hi[STACK][RPT_WAVE];
    __device__ __forceinline__ Stack24 column(uint32_t lane) { return Stack24{&lo[0][lane], &hi[0][lane]}; }
};

struct HitRecord {
    float t;
    uint32_t tri;     /* HIT_MISS or triangle index | backface << 31 */
};

/* intersection.rs:104-122 — NaN-ignoring min/max, strict comparisons as written.  The reference returns
 * tmin or +inf; callers only ever compare that value, so it is kept as (hit, tmin): a hit has a non-NaN
 * tmin < prev_min_t < inf, and "dl > dr" on the inf-encoded values (intersection.rs:216) is
 * hit_r && (!hit_l || tmin_l > tmin_r) — predicates that stay in scalar mask registers. */
template <bool FAST>
__device__ __forceinline__ bool slab_test(float4 lo, float4 hi, F3 ro, F3 rd, F3 ird, float prev_min_t, float &tmin_out) {
    float tx1, tx2, ty1, ty2, tz1, tz2;
    if (FAST) {
        tx1 = rptm::div_by_rcp(lo.x - ro.x, rd.x, ird.x); tx2 = rptm::div_by_rcp(hi.x - ro.x, rd.x, ird.x);
        ty1 = rptm::div_by_rcp(lo.y - ro.y, rd.y, ird.y); ty2 = rptm::div_by_rcp(hi.y - ro.y, rd.y, ird.y);
        tz1 = rptm::div_by_rcp(lo.z - ro.z, rd.z, ird.z); tz2 = rptm::div_by_rcp(hi.z - ro.z, rd.z, ird.z);
    } else {
        tx1 = (lo.x - ro.x) / rd.x; tx2 = (hi.x - ro.x) / rd.x;
        ty1 = (lo.y - ro.y) / rd.y; ty2 = (hi.y - ro.y) / rd.y;
        tz1 = (lo.z - ro.z) / rd.z; tz2 = (hi.z - ro.z) / rd.z;
    }
    float tmin = rptm::fminr(tx1, tx2);
    float tmax = rptm::fmaxr(tx1, tx2);
    tmin = rptm::fmaxr(tmin, rptm::fminr(ty1, ty2));
    tmax = rptm::fminr(tmax, rptm::fmaxr(ty1, ty2));
    tmin = rptm::fmaxr(tmin, rptm::fminr(tz1, tz2));
    tmax = rptm::fminr(tmax, rptm::fmaxr(tz1, tz2));
    tmin_out = tmin;
    return tmax >= tmin && tmax > 0.0f && tmin < prev_min_t;
}

/* How the generic loop reads the scene: the uploaded node array as is (children adjacent, one visit = 64
 * contiguous bytes = half a cache line through L1/L2) and the (a, e1, e2) triangle records.  A finished lane
 * carries count = 0x80000000 so that "at an inner node" / "at a leaf" are single compares on the register (a
 * ballot of a compare is the compare itself; a ballot of a loop-carried bool costs two more VALU instructions). */
template <bool COOP>
struct SceneViewGlobalT {
    static constexpr bool kCoopLeaves = COOP;               /* leaves may hold dozens of triangles: see walk_run.  The streamed walks are
                                                               built both ways and the host picks by the scene's largest leaf: the cooperative
                                                               leaf code costs registers the walk of a thin-leaf scene (every shipped one) needs */
    static constexpr bool kUniformScalar = false;
    const float4 *nodes;
    const float *tri_isect;
    typedef uint2 Cur;                                      /* x = triangle_count, y = left child / first triangle */
    __device__ __forceinline__ Cur root() const { return make_uint2(__float_as_uint(nodes[0].w), __float_as_uint(nodes[1].w)); }
    __device__ __forceinline__ static bool is_inner(Cur c) { return c.x == 0u; }
    __device__ __forceinline__ static bool is_leaf(Cur c) { return (int32_t)c.x > 0; }
    __device__ __forceinline__ static Cur dead() { return make_uint2(0x80000000u, 0u); }
    __device__ __forceinline__ static uint32_t leaf_count(Cur c) { return c.x; }
    __device__ __forceinline__ static uint32_t leaf_first(Cur c) { return c.y; }
    __device__ __forceinline__ void children(Cur c, float4 &lmin, float4 &lmax, float4 &rmin, float4 &rmax) const {
        const float4 *ch = nodes + 2u * c.y;
        lmin = ch[0]; lmax = ch[1]; rmin = ch[2]; rmax = ch[3];
    }
    __device__ __forceinline__ static Cur enter(bool right, float4 lmin, float4 lmax, float4 rmin, float4 rmax) {
        return make_uint2(__float_as_uint(right ? rmin.w : lmin.w), __float_as_uint(right ? rmax.w : lmax.w));
    }
    __device__ __forceinline__ uint32_t far_entry(Cur c, bool far_is_left) const { return far_is_left ? c.y : c.y + 1u; }
    __device__ __forceinline__ Cur from_entry(uint32_t e) const {
        return make_uint2(__float_as_uint(nodes[2u * e].w), __float_as_uint(nodes[2u * e + 1u].w));
    }
    __device__ __forceinline__ void edges(uint32_t ti, F3 &e1, F3 &e2) const {
        const float *p = tri_isect + 9u * (size_t)ti;
        e1 = f3(p[0], p[1], p[2]); e2 = f3(p[3], p[4], p[5]);
    }
    __device__ __forceinline__ F3 corner(uint32_t ti) const {
        const float *p = tri_isect + 9u * (size_t)ti + 6u;
        return f3(p[0], p[1], p[2]);
    }
};
typedef SceneViewGlobalT<true> SceneViewGlobal;

/* The streamed global-memory walks read a PAIR array instead (round 4).  Counters first (profiles/r04_*_pmc_ta.txt): these walks keep the CU's
 * texture-address unit busy 83-92 % of the time (TA_TA_BUSY / TCP_GATE_EN1: VeachMIS shadow 92 %, PBRTest nearest 91 %; the LDS walk 12 %) — that
 * front end is what bounds them.  What a load costs it (tools/microbench/ta_rates.hip, profiles/r04_ta_rates.txt): ~0.5 cycles per LIVE lane and
 * ~10 per instruction whatever the width when the lanes diverge, the data return (64 bytes per clock) on top where lanes share lines.  A visit was
 * four 16-byte loads per lane — the two 32-byte nodes as uploaded, 48 bytes of boxes and 16 of (count, child / first) words.  Here a child pair is
 * ONE 64-byte-aligned record
 *     q0 = (L.lo.xyz, L.hi.x)  q1 = (L.hi.yz, R.lo.xy)  q2 = (R.lo.z, R.hi.xyz)  [8 bytes unused]  (link L, link R)
 * with link = triangle_count << 24 | left child / first triangle: three 16-byte loads and one 8-byte load (still four instructions: boxes are 48
 * bytes; the 8-byte one returns half the data on the shared lines near the top of the tree), and a popped node index costs one 4-byte load from
 * `links[]` instead of two.  Pair p = the children (2p + 1, 2p + 2) of the reference's node pool (its builder allocates children in pairs after the
 * root); a scene whose pool is not pair-shaped, or with a leaf of 255+ triangles or 2^24+ triangles, keeps the one-shot generic walks.  The node
 * is one register: an inner node is its left child's index (< 2^24). */
#ifndef RPT_GSTREAM_UNIFORM_SCALAR
#define RPT_GSTREAM_UNIFORM_SCALAR 1
#endif
template <bool COOP>
struct SceneViewPairsT {
    static constexpr bool kCoopLeaves = COOP;
    const float4 *pairs;          /* 64 bytes per pair: 3 x float4 of boxes, 8 bytes unused, (link L, link R) in the LAST 8 bytes — at offset 48, 16-byte aligned,
                                     the compiler widens the 8-byte load to a 16-byte one; in an array of their own the links cost large scenes a second line */
    const uint32_t *links;        /* per NODE: what a popped stack entry (a node index) resolves to */
    const float *tri_isect;
    typedef uint32_t Cur;
    __device__ __forceinline__ Cur root() const { return links[0]; }
    __device__ __forceinline__ static bool is_inner(Cur c) { return c < (1u << 24); }
    __device__ __forceinline__ static bool is_leaf(Cur c) { return c + 1u > (1u << 24); }          /* (the dead word wraps to 0) */
    __device__ __forceinline__ static Cur dead() { return 0xffffffffu; }
    __device__ __forceinline__ static uint32_t leaf_count(Cur c) { return c >> 24; }
    __device__ __forceinline__ static uint32_t leaf_first(Cur c) { return c & 0xffffffu; }
    __device__ __forceinline__ void children(Cur c, float4 &lmin, float4 &lmax, float4 &rmin, float4 &rmax) const {
        const float4 *p = pairs + 4u * (c >> 1);
        const float4 q0 = p[0], q1 = p[1], q2 = p[2];
        uint2 lk = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(p) + 56);
        asm volatile("" : "+v"(lk.x), "+v"(lk.y));      /* issued WITH the boxes: left alone the compiler sinks this load behind the slab tests, a second round trip */
        lmin = make_float4(q0.x, q0.y, q0.z, __uint_as_float(lk.x));
        lmax = make_float4(q0.w, q1.x, q1.y, 0.0f);
        rmin = make_float4(q1.z, q1.w, q2.x, __uint_as_float(lk.y));
        rmax = make_float4(q2.y, q2.z, q2.w, 0.0f);
    }
    /* The same record through the SCALAR cache, for a node every participating lane stands on (c is wave-uniform): one s_load_dwordx16 instead
     * of four vector loads — no texture-address cycles at all.  A wave of the first iteration is an 8 x 8 pixel block at one sample index, and at
     * the BASELINE resolutions its 64 camera rays walk the same nodes: 98 % of the inner steps of primary-ray waves are wave-uniform on PBRTest
     * 2048^2 and VeachMIS 1080p (tools/uniform_visit_share.py, profiles/r04_uniform_visit_share.txt).  The wait is inside the asm statement: the
     * compiler's s_waitcnt insertion does not see a load it did not emit.  Destinations are early-clobber ("=&s"): an SMEM destination that overlapped
     * its own base pair would be re-read clobbered if the load were ever replayed (XNACK) — LLVM does the same for its own scalar loads on xnack-any
     * targets; tests/test_scalar_path_isa.py checks the emitted registers. */
    static constexpr bool kUniformScalar = RPT_GSTREAM_UNIFORM_SCALAR != 0;
    __device__ __forceinline__ void children_uniform(uint32_t c, float4 &lmin, float4 &lmax, float4 &rmin, float4 &rmax) const {
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        const float4 *p = pairs + 4u * (c >> 1);
        u32x16 r;
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
        lmin = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[14]));
        lmax = make_float4(__uint_as_float(r[3]), __uint_as_float(r[4]), __uint_as_float(r[5]), 0.0f);
        rmin = make_float4(__uint_as_float(r[6]), __uint_as_float(r[7]), __uint_as_float(r[8]), __uint_as_float(r[15]));
        rmax = make_float4(__uint_as_float(r[9]), __uint_as_float(r[10]), __uint_as_float(r[11]), 0.0f);
    }
    __device__ __forceinline__ static Cur enter(bool right, float4 lmin, float4, float4 rmin, float4) { return __float_as_uint(right ? rmin.w : lmin.w); }
    __device__ __forceinline__ uint32_t far_entry(Cur c, bool far_is_left) const { return far_is_left ? c : c + 1u; }
    __device__ __forceinline__ Cur from_entry(uint32_t e) const { return links[e]; }
    /* the same for a wave-uniform popped index / a wave-uniform triangle: scalar cache */
    __device__ __forceinline__ Cur from_entry_uniform(uint32_t e) const {
        uint32_t r;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(links + e) : "memory");
        return r;
    }
    __device__ __forceinline__ void triangle_uniform(uint32_t ti, F3 &e1, F3 &e2, F3 &a) const {
        typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
        const float *p = tri_isect + 9u * (size_t)ti;
        u32x8 r;
        uint32_t r8;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r), "=&s"(r8) : "s"(p) : "memory");
        e1 = f3(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]));
        e2 = f3(__uint_as_float(r[3]), __uint_as_float(r[4]), __uint_as_float(r[5]));
        a = f3(__uint_as_float(r[6]), __uint_as_float(r[7]), __uint_as_float(r8));
    }
    __device__ __forceinline__ void edges(uint32_t ti, F3 &e1, F3 &e2) const {
        const float *p = tri_isect + 9u * (size_t)ti;
        e1 = f3(p[0], p[1], p[2]); e2 = f3(p[3], p[4], p[5]);
    }
    __device__ __forceinline__ F3 corner(uint32_t ti) const {
        const float *p = tri_isect + 9u * (size_t)ti + 6u;
        return f3(p[0], p[1], p[2]);
    }
};
#ifndef RPT_GSTREAM_PAIRS
#define RPT_GSTREAM_PAIRS 1
#endif
#if RPT_GSTREAM_PAIRS
#define RPT_GSTREAM_VIEW(COOP, sc) SceneViewPairsT<COOP>{(sc).gpairs, (sc).glinks, (sc).tri_isect}
#define RPT_GSTREAM_VIEW_SHADOW(COOP, sc) SceneViewPairsT<COOP>{(sc).gpairs_shadow, (sc).glinks_shadow, (sc).tri_isect}
template <bool COOP> struct GstreamView { typedef SceneViewPairsT<COOP> type; };
#else
#define RPT_GSTREAM_VIEW(COOP, sc) SceneViewGlobalT<COOP>{(sc).nodes, (sc).tri_isect}
#define RPT_GSTREAM_VIEW_SHADOW(COOP, sc) SceneViewGlobalT<COOP>{(sc).nodes, (sc).tri_isect}
template <bool COOP> struct GstreamView { typedef SceneViewGlobalT<COOP> type; };
#endif

/* The LDS-resident image of a small scene, built once at upload (rpt_hip.hip, build_lds_image) and copied into
 * LDS by every workgroup.  Measured on MI355X (tools/microbench/valu_rates.hip, SQ counters in profiles/): the
 * traversal kernel is VALU-ISSUE bound — fma/mul/add issue in ~2 cycles per wave64 instruction, everything else
 * (min/max, compares, selects, integer/address ops) in ~4 — so the image is laid out to delete instructions:
 *
 *   plane records   K_A[p] = (L.lo.k, R.lo.k, L.hi.k, R.hi.k),  K_B[p] = (L.hi.k, R.hi.k, L.lo.k, R.lo.k)
 *                   for axis k = x, y, z and child pair p = nodes (2p+1, 2p+2).  A ray whose direction component is
 *                   positive reads K_A, a negative one K_B (a per-ray base address), so the register quad is always
 *                   (L.near, R.near, L.far, R.far): with lo <= hi and a finite non-zero divisor, RN((lo-o)/d) and
 *                   RN((hi-o)/d) are ordered by the sign of d (RN subtraction and division are monotone), so the
 *                   reference's six f32::min/max per box (intersection.rs:108-117) become one max3 and one min3,
 *                   value for value.  Rays outside the exact-division guard keep the explicit min/max on K_A.
 *   descriptors     D[p] = desc(L) | desc(R) << 16;  desc = pair index (< 0x4000) of an inner child,
 *                   0x8000 | triangle_count << 9 | first_triangle for a leaf; 0x4000 marks a finished lane.
 *                   The 16-bit stack holds descriptors, so a pop is one ds_read_u16 — no node lookup.
 *   triangles       a[] | e1[] | e2[]  (16-byte records, one array each)
 *
 * Every load instruction of a visit addresses "array base + 16 * p": the 16 lanes that ds_read_b128 serves per
 * LDS cycle spread over all 64 banks instead of the 4 bank groups an array-of-nodes layout allows.
 * A node array that is not pair-shaped, has an empty/inverted box, or a leaf of 64+ triangles gets no image and
 * is traversed from global memory by the generic loop. */
struct SceneViewLds {
    static constexpr bool kCoopLeaves = false;
    const float4 *img;
    uint32_t pairs, tris, root_desc;
    __device__ __forceinline__ const float4 *tri_base() const { return img + 6u * pairs + ((pairs + 3u) >> 2); }
    __device__ __forceinline__ void edges(uint32_t ti, F3 &e1, F3 &e2) const {
        const float4 *t = tri_base() + ti;
        e1 = xyz4(t[tris]); e2 = xyz4(t[2u * tris]);
    }
    __device__ __forceinline__ F3 corner(uint32_t ti) const { return xyz4(tri_base()[ti]); }
};

/* intersection.rs:9-54 with edge1/edge2 precomputed at upload; the corner is fetched only by lanes that get
 * past the determinant test */
template <typename View>
__device__ __forceinline__ bool moller_trumbore_view(const View &view, uint32_t ti, F3 ro, F3 rd, float &out_t, bool &backface) {
    F3 edge1, edge2;
    view.edges(ti, edge1, edge2);
    F3 pv = cross3(rd, edge2);
    float det = dot3(edge1, pv);
    backface = (rptm::f2u(det) >> 31) != 0u;
    if (rptm::absr(det) < 1e-6f) return false;
    float inv_det = 1.0f / det;
    F3 tv = ro - view.corner(ti);
    float u = dot3(tv, pv) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    F3 qv = cross3(tv, edge1);
    float v = dot3(rd, qv) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot3(edge2, qv) * inv_det;
    if (t < 0.0f) return false;
    out_t = t;
    return true;
}

/* the same test on a record the caller already holds in registers (the wave-cooperative leaves load a leaf once for all the
 * lanes that wait at it) */
__device__ __forceinline__ bool moller_trumbore_regs(F3 edge1, F3 edge2, F3 corner, F3 ro, F3 rd, float &out_t, bool &backface) {
    F3 pv = cross3(rd, edge2);
    float det = dot3(edge1, pv);
    backface = (rptm::f2u(det) >> 31) != 0u;
    if (rptm::absr(det) < 1e-6f) return false;
    float inv_det = 1.0f / det;
    F3 tv = ro - corner;
    float u = dot3(tv, pv) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    F3 qv = cross3(tv, edge1);
    float v = dot3(rd, qv) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot3(edge2, qv) * inv_det;
    if (t < 0.0f) return false;
    out_t = t;
    return true;
}

/* One ray per lane through the BVH.  Per lane the sequence of box tests, triangle tests and the value of the
 * running best t at each of them is exactly the reference's (intersection.rs:177-234); what is scheduled is
 * WHEN a lane takes its next step.  Each trip of the loop, lanes standing on an inner node take one box step;
 * lanes standing on a leaf WAIT until at least RPT_LEAF_K lanes of the wave are waiting (or nobody is left at
 * an inner node), then the wave issues the triangle body once for all of them.
 * Why: after the first bounce the rays of a wave are incoherent.  A replay of the reference traversal on real
 * DarkCornell bounce rays (tools/traversal_sim.py) gives, in issue slots per ray: classic while-while 148
 * (lanes at a leaf wait for the slowest lane of every round), one-step-per-trip "if-if" 113 (the leaf body,
 * 14 % of the steps, is issued on almost every trip), deferred leaves with K = 12..16: 106; ideal 38.  Measured on MI355X the gain is smaller (LDS/latency share the
 * bill with VALU issue): traverse 23.7 -> 22.4..22.8 ms for K = 8..16, 25.9 ms for K = 64 (= while-while); K = 8 is
 * also the best for VeachMIS.
 * `stack` points at this lane's column of the wave's LDS stack: entry e lives at stack[e * RPT_WAVE]. */
#ifndef RPT_LEAF_K
#define RPT_LEAF_K 8
#endif
#ifndef RPT_LEAF_GREEDY_PCT_GLOBAL
#define RPT_LEAF_GREEDY_PCT_GLOBAL 100   /* global-memory walks: 0 = the RPT_LEAF_K threshold rule; > 0 = one body per trip (see lds_walk_run).
                                            Measured (K = 8 rule / 100 / 60): VeachMIS 5175 / 5506 / 5453 Mrays/s, PBRTest 4915 / 5027 / 5011 */
#endif
#ifndef RPT_COOP_LEAF_MIN
#endif
__device__ __forceinline__ float rpt_readlane(float v, int lane) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane)); }
__device__ __forceinline__ uint32_t rpt_readlane_u(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

/* Everything a ray needs besides (ro, rd, 1/rd) and its stack column: the walk can be stopped after a number of loop
 * trips and resumed (the streamed kernels hand finished lanes new rays in between). */
template <typename View> struct Walk {
    typename View::Cur cur;    /* node the ray stands on; View::dead() when finished / no ray */
    int sp;
    HitRecord res;
};
template <typename View>
__device__ __forceinline__ void walk_begin(const View &view, Walk<View> &w) {
    w.cur = view.root();
    w.sp = 0;
    w.res.t = 1000000.0f;
    w.res.tri = HIT_MISS;
}
template <typename View>
__device__ __forceinline__ bool walk_dead(const Walk<View> &w) { return !View::is_inner(w.cur) && !View::is_leaf(w.cur); }

/* At most `budget` trips of the deferred-leaf loop for the lanes of this wave; returns early when no lane has anything
 * left.  Per ray the visiting order and every comparison are the reference's. */
/* the node behind a popped stack entry; a wave-uniform entry (coherent camera rays pop together) comes through the scalar cache */
template <bool ANY_HIT, typename View>
__device__ __forceinline__ typename View::Cur walk_pop(const View &view, uint32_t e) {
    if constexpr (View::kUniformScalar && !ANY_HIT && !View::kCoopLeaves) {      /* (measured: + 1.5 % PBRTest, + 0.6 % VeachMIS; nothing on any-hit walks and on the fat-leaf build) */
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
        if (rpt_ballot(e == e0) == rpt_ballot(true)) return view.from_entry_uniform(e0);
        asm volatile("" ::: "memory");
        return view.from_entry(e);
    } else {
        return view.from_entry(e);
    }
}

template <int STACK, bool ANY_HIT, bool FAST, bool FIXED = false, typename View, typename StackRef>
__device__ __forceinline__ void walk_run(const View &view, Walk<View> &w, F3 ro, F3 rd, F3 ird, float max_t, StackRef &stack, int budget) {
    static_assert(!FIXED || ANY_HIT, "only the any-hit walk may choose its order");
    typedef typename View::Cur Cur;
    HitRecord res = w.res;
    int sp = w.sp;
    Cur cur = w.cur;
    for (int trip = 0; trip < budget; ++trip) {
        const bool at_inner = View::is_inner(cur);
        const bool at_leaf = View::is_leaf(cur);
        const unsigned long long inner_m = rpt_ballot(at_inner), leaf_m = rpt_ballot(at_leaf);
        if ((inner_m | leaf_m) == 0ull) break;
#if RPT_LEAF_GREEDY_PCT_GLOBAL
        if (at_inner && !((uint32_t)__popcll(leaf_m) * 100u > (uint32_t)__popcll(inner_m) * (uint32_t)RPT_LEAF_GREEDY_PCT_GLOBAL)) {
#else
        if (at_inner) {
#endif
            /* inner node (:207-229): test both children against the current best t */
            float4 lmin, lmax, rmin, rmax;
            float tl, tr;
            bool hit_l, hit_r;
            if constexpr (View::kUniformScalar) {
                /* all the lanes of this step on ONE node (a wave of camera rays: nearly always): its record comes through the scalar cache, and the
                 * slab tests read the planes as scalar operands (tested INSIDE the branch: merged behind it, fourteen v_mov would carry them into VGPRs) */
                const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                if (rpt_ballot(cur == c0) == rpt_ballot(true)) {
                    view.children_uniform(c0, lmin, lmax, rmin, rmax);
                    hit_l = slab_test<FAST>(lmin, lmax, ro, rd, ird, res.t, tl);
                    hit_r = slab_test<FAST>(rmin, rmax, ro, rd, ird, res.t, tr);
                    asm volatile("" : "+v"(tl), "+v"(tr));      /* (or the optimiser sinks both branches' tests into ONE copy behind the branch) */
                } else {
                    asm volatile("" ::: "memory");      /* (keeps the four vector loads on THIS side of the branch: hoisted above it they are issued on every step) */
                    view.children(cur, lmin, lmax, rmin, rmax);
                    hit_l = slab_test<FAST>(lmin, lmax, ro, rd, ird, res.t, tl);
                    hit_r = slab_test<FAST>(rmin, rmax, ro, rd, ird, res.t, tr);
                }
            } else {
                view.children(cur, lmin, lmax, rmin, rmax);
                hit_l = slab_test<FAST>(lmin, lmax, ro, rd, ird, res.t, tl);
                hit_r = slab_test<FAST>(rmin, rmax, ro, rd, ird, res.t, tr);
            }
            const bool swap = FIXED ? (hit_r && !hit_l) : (hit_r && (!hit_l || tl > tr));     /* strict: ties keep left first */
            if (hit_l || hit_r) {
                if (hit_l && hit_r && sp < STACK) {
                    stack_put(stack, sp, view.far_entry(cur, swap));
                    sp += 1;
                }
                cur = View::enter(swap, lmin, lmax, rmin, rmax);
            } else if (sp == 0) {
                cur = View::dead();
            } else {
                sp -= 1;
                cur = walk_pop<ANY_HIT>(view, stack_get(stack, sp));
            }
        }
#if RPT_LEAF_GREEDY_PCT_GLOBAL
        const bool do_leaf = (uint32_t)__popcll(leaf_m) * 100u > (uint32_t)__popcll(inner_m) * (uint32_t)RPT_LEAF_GREEDY_PCT_GLOBAL;
#else
        const bool do_leaf = (uint32_t)__popcll(leaf_m) >= (uint32_t)RPT_LEAF_K || inner_m == 0ull;
#endif
        if (do_leaf) {                                                                       /* (wave-uniform) */
            bool accepted = false, coop_done = false;
            const uint32_t count = View::leaf_count(cur), first = View::leaf_first(cur);
            if constexpr (View::kCoopLeaves) {
                /* FAT leaves, wave-cooperatively.  The reference's builder stops splitting where the SAH says so, and on
                 * clustered geometry that leaves up to 64 triangles in a leaf (the 1 M-triangle stand-in: 140 triangle
                 * tests per ray).  One lane looping over 64 triangles while the other 63 wait ran that scene at 7 % lane
                 * utilisation (profiles/r02base_deepbvh_pmc_sq.txt).  Instead the owner's ray is broadcast (readlane:
                 * it lives in scalar registers) and every lane tests ONE triangle of the leaf.  The sequential loop
                 * accepts t_i < running best in index order, i.e. ends with the smallest t and, among equal t, the lowest
                 * index (any-hit: the lowest index that passes) — which is what the scalar scan below selects. */
                const bool fat = at_leaf && count > (uint32_t)RPT_COOP_LEAF_MIN;
                unsigned long long todo = rpt_ballot(fat);
                if (todo != 0ull) {
                    const unsigned long long exec_m = rpt_ballot(true);
                    const uint32_t n_act = (uint32_t)__popcll(exec_m);
                    const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(exec_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)exec_m, 0u));
                    coop_done = fat;
                    do {
                        /* ONE load of the leaf's records serves every lane of the wave that waits at this very leaf: a wave is one
                         * 8 x 8 pixel block, so after generation (and for shadow rays towards one light) most of a wave
                         * stands on the same leaf — each used to fetch the 2.3 KB again */
                        const int lead = __ffsll((long long)todo) - 1;
                        const uint32_t b_count = rpt_readlane_u(count, lead), b_first = rpt_readlane_u(first, lead);
                        const unsigned long long group = rpt_ballot(fat && first == b_first && count == b_count) & todo;
                        todo &= ~group;
                        for (uint32_t base = 0; base < b_count; base += n_act) {
                            const bool mine = base + my_rank < b_count;
                            const uint32_t ti = b_first + base + my_rank;
                            F3 e1 = f3(0, 0, 0), e2 = f3(0, 0, 0), corner = f3(0, 0, 0);
                            if (mine) {
                                view.edges(ti, e1, e2);
                                corner = view.corner(ti);
                            }
                            unsigned long long g = ANY_HIT ? (group & ~rpt_ballot(accepted)) : group;
                            while (g != 0ull) {
                                const int L = __ffsll((long long)g) - 1;
                                g &= g - 1ull;
                                const F3 bo = f3(rpt_readlane(ro.x, L), rpt_readlane(ro.y, L), rpt_readlane(ro.z, L));
                                const F3 bd = f3(rpt_readlane(rd.x, L), rpt_readlane(rd.y, L), rpt_readlane(rd.z, L));
                                const float b_max = ANY_HIT ? rpt_readlane(max_t, L) : 0.0f;
                                uint32_t best_bits = __float_as_uint(rpt_readlane(res.t, L));      /* positive floats order like their bits */
                                uint32_t best_tri = HIT_MISS;
                                float t = 0.0f;
                                bool bf = false;
                                const bool acc = mine && moller_trumbore_regs(e1, e2, corner, bo, bd, t, bf) && t > 0.001f &&
                                                 __float_as_uint(t) < best_bits && (!ANY_HIT || t <= b_max);
                                unsigned long long am = rpt_ballot(acc);
                                while (am != 0ull) {                                          /* scalar scan, lowest triangle first */
                                    const int l = __ffsll((long long)am) - 1;
                                    am &= am - 1ull;
                                    const uint32_t tb = __float_as_uint(rpt_readlane(t, l));
                                    if (tb < best_bits) {
                                        best_bits = tb;
                                        best_tri = rpt_readlane_u(ti, l) | (rpt_readlane_u(bf ? 1u : 0u, l) << 31);
                                        if (ANY_HIT) break;
                                    }
                                }
                                if ((int)__lane_id() == L && best_tri != HIT_MISS) {
                                    res.t = __uint_as_float(best_bits);
                                    res.tri = best_tri;
                                    accepted = true;
                                }
                            }
                        }
                    } while (todo != 0ull);
                }
            }
            if (at_leaf && !coop_done) {
                /* leaf triangles in index order (:186-205) */
                bool leaf_uniform = false;
                uint32_t c0 = 0u;
                if constexpr (View::kUniformScalar && !ANY_HIT && !View::kCoopLeaves) {
                    /* every lane of this step on ONE leaf (coherent camera rays): its 36-byte triangle records through the scalar cache */
                    c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                    leaf_uniform = rpt_ballot(cur == c0) == rpt_ballot(true);
                }
                if constexpr (View::kUniformScalar && !ANY_HIT && !View::kCoopLeaves) {
                    if (leaf_uniform) {
                        const uint32_t n0 = View::leaf_count(c0), f0 = View::leaf_first(c0);
                        for (uint32_t i = 0; i < n0; ++i) {
                            const uint32_t ti = f0 + i;
                            F3 e1, e2, a;
                            view.triangle_uniform(ti, e1, e2, a);
                            float t = 0.0f;
                            bool bf = false;
                            if (moller_trumbore_regs(e1, e2, a, ro, rd, t, bf) && t > 0.001f && t < res.t) {
                                asm volatile("" ::: "memory");
                                res.t = t;
                                res.tri = ti | (bf ? 0x80000000u : 0u);
                            }
                        }
                    }
                }
                if (!leaf_uniform) {
                asm volatile("" ::: "memory");
                for (uint32_t i = 0; i < count; ++i) {
                    uint32_t ti = first + i;
                    float t = 0.0f;
                    bool bf = false;
                    if (moller_trumbore_view(view, ti, ro, rd, t, bf) && t > 0.001f && t < res.t && (!ANY_HIT || t <= max_t)) {
                        /* result.t = result.t.min(t) with t < result.t already established (intersection.rs:195-199).  Kept a
                         * real branch: as two selects on vcc the update becomes back-to-back VOP2 v_cndmask, which gfx950 issues
                         * at ~22 cycles each (tools/microbench/valu_rates.hip) */
                        asm volatile("" ::: "memory");
                        res.t = t;
                        res.tri = ti | (bf ? 0x80000000u : 0u);
                        if (ANY_HIT) { accepted = true; break; }
                    }
                }
                }
            }
            if (at_leaf) {
                if ((ANY_HIT && accepted) || sp == 0) {
                    cur = View::dead();
                } else {
                    sp -= 1;
                    cur = walk_pop<ANY_HIT>(view, stack_get(stack, sp));
                }
            }
        }
    }
    w.cur = cur;
    w.sp = sp;
    w.res = res;
}

template <int STACK, bool ANY_HIT, bool FAST, bool FIXED = false, typename View, typename StackRef>
__device__ __forceinline__ HitRecord traverse_loop(const View &view, F3 ro, F3 rd, F3 ird, float max_t, StackRef &stack) {
    Walk<View> w;
    walk_begin(view, w);
    walk_run<STACK, ANY_HIT, FAST, FIXED>(view, w, ro, rd, ird, max_t, stack, 0x7fffffff);
    return w.res;
}

template <int STACK, bool ANY_HIT, typename View, typename StackT>
__device__ __forceinline__ HitRecord traverse_one(const View &view, uint32_t fastdiv_ok, F3 ro, F3 rd, float max_t, StackT *stack) {
    bool fast = fastdiv_ok != 0u && rptm::fastdiv_divisor_ok(rd.x) && rptm::fastdiv_divisor_ok(rd.y) && rptm::fastdiv_divisor_ok(rd.z) &&
                rptm::fastdiv_operand_ok(ro.x) && rptm::fastdiv_operand_ok(ro.y) && rptm::fastdiv_operand_ok(ro.z);
    if (fast) {
        F3 ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        return traverse_loop<STACK, ANY_HIT, true>(view, ro, rd, ird, max_t, stack);
    }
    return traverse_loop<STACK, ANY_HIT, false>(view, ro, rd, rd, max_t, stack);
}

/* The same walk over the LDS image (SceneViewLds).  SIGNED = the ray passed the exact-division guard: plane
 * records are read through the per-ray sign-selected bases and near/far need no min/max. */
template <bool SIGNED>
__device__ __forceinline__ bool slab_pair_lds(float n_x, float n_y, float n_z, float f_x, float f_y, float f_z, F3 ro, F3 rd, F3 ird,
                                              float prev_min_t, float &tmin_out) {
    float tmin, tmax;
    if (SIGNED) {
        float a = rptm::div_by_rcp(n_x - ro.x, rd.x, ird.x), b = rptm::div_by_rcp(n_y - ro.y, rd.y, ird.y), c = rptm::div_by_rcp(n_z - ro.z, rd.z, ird.z);
        float d = rptm::div_by_rcp(f_x - ro.x, rd.x, ird.x), e = rptm::div_by_rcp(f_y - ro.y, rd.y, ird.y), f = rptm::div_by_rcp(f_z - ro.z, rd.z, ird.z);
        tmin = __builtin_fmaxf(__builtin_fmaxf(a, b), c);       /* no NaN on this path: plain max3 / min3 */
        tmax = __builtin_fminf(__builtin_fminf(d, e), f);
    } else {
        float tx1 = (n_x - ro.x) / rd.x, tx2 = (f_x - ro.x) / rd.x;
        float ty1 = (n_y - ro.y) / rd.y, ty2 = (f_y - ro.y) / rd.y;
        float tz1 = (n_z - ro.z) / rd.z, tz2 = (f_z - ro.z) / rd.z;
        tmin = rptm::fminr(tx1, tx2);
        tmax = rptm::fmaxr(tx1, tx2);
        tmin = rptm::fmaxr(tmin, rptm::fminr(ty1, ty2));
        tmax = rptm::fminr(tmax, rptm::fmaxr(ty1, ty2));
        tmin = rptm::fmaxr(tmin, rptm::fminr(tz1, tz2));
        tmax = rptm::fminr(tmax, rptm::fmaxr(tz1, tz2));
    }
    tmin_out = tmin;
    return tmax >= tmin && tmax > 0.0f && tmin < prev_min_t;
}

/* leaf batching threshold of the LDS loop: its inner step is ~25 % cheaper than the generic one, so waiting for
 * more leaf lanes pays (one ray per lane: traverse 28.4 ms at K = 8, 26.8-27.3 ms for K = 16..32, 29.0 ms at 48;
 * streamed: 26.0 / 23.7 / 23.5 / 22.6 / 23.3 ms for K = 8 / 12 / 16 / 24 / 32) */
#ifndef RPT_LEAF_K_LDS
#define RPT_LEAF_K_LDS 24
#endif
#ifndef RPT_LEAF_GREEDY_PCT
#define RPT_LEAF_GREEDY_PCT 100    /* 0: the threshold rule above; > 0: one body per trip, see lds_walk_run (measured, streamed DarkCornell:
                                      traverse 82.6 / 80.0 / 81.4 / 82.8 ms per 8 batches for 0 / 100 / 130 / 170) */
#endif
/* A walk over the LDS image that can be stopped after a number of loop trips and resumed (k_traverse_nearest_stream):
 * everything a ray needs besides (ro, rd, 1/rd) is in here and in its stack column. */
struct LdsWalk {
    uint32_t cur;          /* descriptor of the node the ray stands on; LDS_DESC_DEAD when finished / no ray */
    int sp;
    HitRecord res;
};
__device__ __forceinline__ void lds_walk_begin(const SceneViewLds &view, LdsWalk &w) {
    w.cur = view.root_desc;
    w.sp = 0;
    w.res.t = 1000000.0f;
    w.res.tri = HIT_MISS;
}

/* At most `budget` trips of the deferred-leaf loop (see traverse_loop) for the lanes of this wave; returns early when
 * no lane has anything left.  Per ray the visiting order and every comparison are the reference's. */
/* MIXED (the last extension rays of a batch without NEE, k_traverse_nearest_stream<.., LAST = true>): a nearest-hit walk in which SOME lanes only have to
 * answer "hit or miss" — `stop_first` lanes leave at their first accepted triangle, which is where the reference's walk makes result.hit true for good
 * (intersection.rs:195-203), and because nothing was accepted before, result.t was 1e6 at every box test up to there: the part of the walk they run is an
 * any-hit walk, whose answer does not depend on the visiting order (header).  They read the planes and child descriptors of `img_lane` — the flipped copy
 * of the pair records when the workgroup staged one, then with order_bias = +inf (tl > tr + inf is never true: the left child first, the fixed order of
 * shadow_order.h choose_last_order); the other lanes read the primary image with order_bias = 0 (tr + 0 compares like tr) and are the reference's walk to its end. */
/* PRESUB (the camera rays of a call's first iteration, k_traverse_nearest_stream FIRST): every ray of the launch has the SAME origin, and the workgroup
 * staged the plane records with that origin already subtracted — the very `plane - ro` (one IEEE subtraction of the same two floats) each lane would
 * compute at each of the twelve planes of a node pair.  The slab test then divides the staged value directly; the triangle test keeps the true origin. */
template <int STACK, bool ANY_HIT, bool SIGNED, bool FIXED = false, bool MIXED = false, bool PRESUB = false>
__device__ __forceinline__ void lds_walk_run(const SceneViewLds &view, LdsWalk &w, F3 ro, F3 rd, F3 ird, float max_t, uint16_t *stack,
                                             int budget, const float4 *img_lane = nullptr, uint32_t stop_first = 0u, float order_bias = 0.0f) {
    static_assert(!FIXED || ANY_HIT, "only the any-hit walk may choose its order");
    static_assert(!MIXED || (!ANY_HIT && !FIXED), "MIXED is the nearest-hit walk with per-lane early exits");
    const F3 ro_slab = PRESUB ? f3(0.0f, 0.0f, 0.0f) : ro;         /* x - (+0) is x, bit for bit: the subtraction folds away */
    const uint32_t P = view.pairs;
    const float4 *img = MIXED ? img_lane : view.img;
    /* per-ray plane-record bases (float4 units): x | y | z, A or B variant by the sign of the direction */
    const float4 *px = img + ((SIGNED && rd.x < 0.0f) ? P : 0u);
    const float4 *py = img + 2u * P + ((SIGNED && rd.y < 0.0f) ? P : 0u);
    const float4 *pz = img + 4u * P + ((SIGNED && rd.z < 0.0f) ? P : 0u);
    const uint32_t *descs = reinterpret_cast<const uint32_t *>(img + 6u * P);
    uint32_t cur = w.cur;
    int sp = w.sp;
    HitRecord res = w.res;
    for (int trip = 0; trip < budget; ++trip) {
        const bool at_inner = cur < LDS_DESC_DEAD;
        const bool at_leaf = cur >= LDS_DESC_LEAF;
        const unsigned long long inner_m = rpt_ballot(at_inner), leaf_m = rpt_ballot(at_leaf);
        if ((inner_m | leaf_m) == 0ull) break;
#if RPT_LEAF_GREEDY_PCT
        /* ONE body per trip, the one with more lanes ready for it (a leaf step counts RPT_LEAF_GREEDY_PCT % of an inner one):
         * lanes on a leaf no longer sit out a fixed quota of inner steps, and no body is issued for a handful of lanes */
        /* (the compiler evaluates this wave-uniform comparison on the vector unit, v_mov + v_cmp_gt_u64 per trip; forced into
         * scalar registers with s_cmp / s_cselect the kernel got SLOWER, 75.8 -> 77.0 ms: the scalar chain bcnt -> mul -> cmp ->
         * cselect -> nor -> saveexec is latency the vector form hides) */
        const bool do_leaf = (uint32_t)__popcll(leaf_m) * 100u > (uint32_t)__popcll(inner_m) * (uint32_t)RPT_LEAF_GREEDY_PCT;
        if (at_inner && !do_leaf) {
#else
        if (at_inner) {
#endif
            const float4 X = px[cur], Y = py[cur], Z = pz[cur];     /* (L.near, R.near, L.far, R.far) per axis */
            const uint32_t d = descs[cur];
            float tl, tr;
            const bool hit_l = slab_pair_lds<SIGNED>(X.x, Y.x, Z.x, X.z, Y.z, Z.z, ro_slab, rd, ird, res.t, tl);
            const bool hit_r = slab_pair_lds<SIGNED>(X.y, Y.y, Z.y, X.w, Y.w, Z.w, ro_slab, rd, ird, res.t, tr);
            const bool swap = FIXED ? (hit_r && !hit_l)
                            : MIXED ? (hit_r && (!hit_l || tl > tr + order_bias))
                                    : (hit_r && (!hit_l || tl > tr));     /* strict: ties keep left first */
            if (hit_l || hit_r) {
                const uint32_t nf = __builtin_amdgcn_alignbit(d, d, swap ? 16u : 0u);    /* near | far << 16 */
                if (hit_l && hit_r && sp < STACK) {
                    stack[sp * RPT_WAVE] = (uint16_t)(nf >> 16);
                    sp += 1;
                }
                cur = nf & 0xffffu;
            } else if (sp == 0) {
                cur = LDS_DESC_DEAD;
            } else {
                sp -= 1;
                cur = stack[sp * RPT_WAVE];
            }
        }
#if RPT_LEAF_GREEDY_PCT
        if (at_leaf && do_leaf) {
#else
        if (at_leaf && ((uint32_t)__popcll(leaf_m) >= (uint32_t)RPT_LEAF_K_LDS || inner_m == 0ull)) {
#endif
            bool accepted = false;
            const uint32_t count = (cur >> 9) & 63u, first = cur & 511u;
            for (uint32_t i = 0; i < count; ++i) {
                uint32_t ti = first + i;
                float t = 0.0f;
                bool bf = false;
                if (moller_trumbore_view(view, ti, ro, rd, t, bf) && t > 0.001f && t < res.t && (!ANY_HIT || t <= max_t)) {
                    /* result.t = result.t.min(t) with t < result.t already established (intersection.rs:195-199).  Kept a
                     * real branch: as two selects on vcc the update becomes back-to-back VOP2 v_cndmask, which gfx950 issues
                     * at ~22 cycles each (tools/microbench/valu_rates.hip) */
                    asm volatile("" ::: "memory");
                    res.t = t;
                    res.tri = ti | (bf ? 0x80000000u : 0u);
                    if (ANY_HIT || (MIXED && stop_first != 0u)) { accepted = true; break; }
                }
            }
            if (((ANY_HIT || MIXED) && accepted) || sp == 0) {
                cur = LDS_DESC_DEAD;
            } else {
                sp -= 1;
                cur = stack[sp * RPT_WAVE];
            }
        }
    }
    w.cur = cur;
    w.sp = sp;
    w.res = res;
}

template <int STACK, bool ANY_HIT, bool SIGNED, bool FIXED = false>
__device__ __forceinline__ HitRecord traverse_loop_lds(const SceneViewLds &view, F3 ro, F3 rd, F3 ird, float max_t, uint16_t *stack) {
    LdsWalk w;
    lds_walk_begin(view, w);
    lds_walk_run<STACK, ANY_HIT, SIGNED, FIXED>(view, w, ro, rd, ird, max_t, stack, 0x7fffffff);
    return w.res;
}

__device__ __forceinline__ bool fastdiv_ray_ok(uint32_t fastdiv_ok, F3 ro, F3 rd) {
    return fastdiv_ok != 0u && rptm::fastdiv_divisor_ok(rd.x) && rptm::fastdiv_divisor_ok(rd.y) && rptm::fastdiv_divisor_ok(rd.z) &&
           rptm::fastdiv_operand_ok(ro.x) && rptm::fastdiv_operand_ok(ro.y) && rptm::fastdiv_operand_ok(ro.z);
}

template <int STACK, bool ANY_HIT>
__device__ __forceinline__ HitRecord traverse_one(const SceneViewLds &view, uint32_t fastdiv_ok, F3 ro, F3 rd, float max_t, uint16_t *stack) {
    if (fastdiv_ray_ok(fastdiv_ok, ro, rd)) {
        F3 ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        return traverse_loop_lds<STACK, ANY_HIT, true>(view, ro, rd, ird, max_t, stack);
    }
    return traverse_loop_lds<STACK, ANY_HIT, false>(view, ro, rd, rd, max_t, stack);
}

/* Small scenes live in LDS: when the traversal image (SceneViewLds) fits in RPT_LDS_SCENE_BYTES
 * every workgroup copies the upload-time LDS image in once and traverses out of LDS
 * (ds_read_b128, ~64-cycle latency, no pressure on the CU's single vector-memory address
 * unit — the measured limiter once the divisions were gone: ~380 divergent 16-byte
 * wave-loads per wave through one TA per CU).  Larger scenes read through L1/L2. */
extern __shared__ __attribute__((aligned(16))) float4 rpt_lds_dyn[];   /* sized at launch to the scene (LDS variants only) */
template <bool LDS_SCENE> struct SceneViewOf { typedef SceneViewGlobal type; };
template <> struct SceneViewOf<true> { typedef SceneViewLds type; };

template <int THREADS>
__device__ __forceinline__ SceneViewLds stage_scene_lds(const DevScene &sc, float4 *lds_scene, bool shadow_copy = false) {
    const float4 *image = shadow_copy ? sc.lds_image_shadow : sc.lds_image;       /* (the flipped copy: same sizes, same root) */
    for (uint32_t k = threadIdx.x; k < sc.lds_vecs; k += THREADS) lds_scene[k] = image[k];
    __syncthreads();
    return SceneViewLds{lds_scene, sc.lds_pairs, sc.n_triangles, sc.lds_root};
}
template <bool LDS_SCENE, int THREADS>
__device__ __forceinline__ typename SceneViewOf<LDS_SCENE>::type stage_scene(const DevScene &sc, float4 *lds_scene) {
    if constexpr (LDS_SCENE) return stage_scene_lds<THREADS>(sc, lds_scene);
    else return SceneViewGlobal{sc.nodes, sc.tri_isect};
}

/* Per-iteration bookkeeping that needs no kernel of its own (one thread of the traversal launch).  The shadow queue was
 * consumed by the previous iteration's shadow kernel (same stream).  The sky stage is lazy (k_sky): it drained its queue last
 * iteration only if enough misses had piled up or nothing else was left — the same decision is re-derived here from the same,
 * still unmodified words. */
__device__ __forceinline__ void iteration_bookkeeping(const DevQueues &q, uint32_t iteration) {
    const uint32_t prev = (iteration + 1u) & 1u;
    uint32_t positions, waiting;
    q_extent(q.sky_cnt, positions, waiting);
    q_clear(q.shadow_cnt);
    if (q.sky_at_end == 0u && (waiting >= q.sky_threshold || q.count[Q_ALIVE0 + prev * Q_LINE] == 0u)) q_clear(q.sky_cnt);
    q.count[Q_ALIVE0 + prev * Q_LINE] = 0u;
    q.count[Q_REGEN0 + prev * Q_LINE] = 0u;
}

/* Extension rays.  Thread i owns slot i; it traces the slot's ray if one is
 * pending (HIT_PENDING) and writes the hit record into hit[slot].  A wave that
 * found work raises this iteration's alive flag (plain store, every writer
 * stores the same value), which the shade stage reports to the host. */
template <int STACK, bool LDS_SCENE, int THREADS, bool SMALL = false>
__global__ __launch_bounds__(THREADS) void k_traverse_nearest(DevScene sc, DevState st, DevQueues q, uint32_t iteration) {
    /* LDS-resident scenes walk 16-bit descriptors: 16-bit stack entries (32 KB per 1024-thread workgroup, which with a
     * <= 32 KB scene image is the 64 KB a workgroup may hold: 2 workgroups = 32 waves per CU) */
    typedef typename StackElem<LDS_SCENE || SMALL>::type StackT;
    __shared__ StackT lds_stack[THREADS / RPT_WAVE][STACK][RPT_WAVE];
    float4 *lds_scene = rpt_lds_dyn;
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    const uint32_t slot = blockIdx.x * THREADS + threadIdx.x;
    if (slot == 0u) {
        iteration_bookkeeping(q, iteration);
    }
    bool pending = false;
    if (slot < st.n_slots) pending = __float_as_uint(st.hit[slot].y) == HIT_PENDING;
    if (LDS_SCENE && !__syncthreads_or(pending)) return;      /* block-uniform: nothing to trace here */
    const auto view = stage_scene<LDS_SCENE, THREADS>(sc, lds_scene);
    unsigned long long active = rpt_ballot(pending);
    if (active == 0ull) return;
    if (__lane_id() == (uint32_t)__ffsll((long long)active) - 1u) {
        raise_flag(&q.count[Q_ALIVE0 + (iteration & 1u) * Q_LINE]);
        /* ray accounting: sharded, non-returning atomics (nobody waits for them) */
        atomicAdd(&q.ray_shards[(blockIdx.x % RPT_STAT_SHARDS) * RPT_STAT_STRIDE], (unsigned long long)__popcll(active));
    }
    if (!pending) return;
    float4 ra = st.ray_a[slot];
    float2 rb = st.ray_b[slot];
    F3 ro = f3(ra.x, ra.y, ra.z), rd = f3(ra.w, rb.x, rb.y);
    StackT *stack = &lds_stack[threadIdx.x / RPT_WAVE][0][threadIdx.x % RPT_WAVE];
    HitRecord h = traverse_one<STACK, false>(view, sc.fastdiv_ok, ro, rd, 0.0f, stack);
    st.hit[slot] = make_float2(h.t, __uint_as_float(h.tri));
}

/* Extension rays of an LDS-resident scene, STREAMED: a workgroup owns up to RPT_STREAM_RAYS x THREADS consecutive slots
 * and deals them to the idle lanes of its waves on demand.
 * The traversal is VALU-issue bound and after the first bounce the rays of a wave need very different numbers of
 * trips (DarkCornell bounce 2: median 25, p90 34, max 68 node visits), so a one-ray-per-lane wave spends most of its
 * trips with a minority of lanes alive (lane utilisation 40 %).  Here, every RPT_STREAM_TRIPS trips the wave looks at
 * its idle lanes; when at least RPT_STREAM_REFILL are idle they write their hit records and take the next slots from the
 * workgroup's pool (an LDS counter over the workgroup's RPT_STREAM_RAYS x THREADS consecutive slots).  The walk itself (lds_walk_run) is the same code with a trip budget: no per-lane bookkeeping inside
 * the hot loop.  Per ray nothing changes — same tests in the same order — so hit records are the reference's bit for
 * bit, and slots stay identity mapped (a slot's ray is traced by SOME lane of the wave that owns its range). */
#ifndef RPT_STREAM_RAYS
#define RPT_STREAM_RAYS 8          /* most slots per lane of a workgroup (the host lowers it for small launches) */
#endif
/* a wave looks for new rays every RPT_STREAM_TRIPS loop trips, once RPT_STREAM_REFILL of its lanes are idle.  Re-measured with 64
 * pixels per wave (round 3, three boxes, DarkCornell Mrays/s relative to 8 / 12): trips 4 / 12 / 16 / 24 / 32: -2.3 / +0.4 / +0.8 /
 * +1.1 / -0.2 %; refill 8 / 16 / 24 at 8 trips: -1 / +-0 / +-0; 16 / 16: +1.1 ... +1.7 % (and +1.1 % with nee = MIS, +1.0 % on 1/8 of
 * the image); 20 / 16 and 24 / 16 the same within noise, 16 / 20 less. */
#ifndef RPT_STREAM_TRIPS
#define RPT_STREAM_TRIPS 16
#endif
#ifndef RPT_STREAM_REFILL
#define RPT_STREAM_REFILL 16
#endif
/* The workgroup's pool is one 64-bit LDS word (next slot | end slot << 32): a wave takes slots with ONE 64-bit ds_add that
 * returns a consistent (next, end) pair.  When the span is used up the wave that notices fetches the next span of SPAN
 * slots from the launch-wide counter (one global atomic per SPAN slots) — PERSISTENT workgroups: the grid holds as many
 * workgroups as the GPU keeps resident, and none of them drains before the whole launch runs out of slots.  (Round 1
 * gave every workgroup one fixed span: each of the 4 096 workgroups then ended in its own tail of ever emptier waves —
 * the replay, tools/traversal_sim.py, puts 17 % of the issue slots there — and the launch in a tail of late workgroups.
 * Stealing 512-slot chunks per WAVE from one global counter was measured slower: 65 k atomics per launch on one address.) */
struct WgPool {
    unsigned long long word;     /* lo = next slot, hi = end of the span; hi == 0: the launch has no slots left */
    uint32_t lock;
};
/* one lane: take up to `want` slots.  Returns the first slot and how many were obtained (0: none right now);
 * *finished is set once the launch-wide pool is empty. */
__device__ __forceinline__ uint32_t wg_pool_take(WgPool *pool, uint32_t *global_next, uint32_t n_slots, uint32_t SPAN, uint32_t want,
                                                 uint32_t &got, bool &finished) {
    got = 0u;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const unsigned long long v = atomicAdd(&pool->word, (unsigned long long)want);
        const uint32_t next = (uint32_t)v, end = (uint32_t)(v >> 32);
        if (next < end) {
            got = end - next < want ? end - next : want;
            return next;
        }
        if (end == 0u) { finished = true; return 0u; }
        if (atomicCAS(&pool->lock, 0u, 1u) != 0u) return 0u;            /* another wave is fetching the next span: look again later */
        const unsigned long long now = atomicAdd(&pool->word, 0ull);
        if ((uint32_t)now >= (uint32_t)(now >> 32) && (uint32_t)(now >> 32) != 0u) {
            const uint32_t g = atomicAdd(global_next, SPAN);
            const unsigned long long fresh = g < n_slots ? ((unsigned long long)(g + SPAN < n_slots ? g + SPAN : n_slots) << 32) | g
                                                         : 0x00000000f0000000ull;
            atomicExch(&pool->word, fresh);
        }
        __threadfence_block();
        atomicExch(&pool->lock, 0u);
    }
    return 0u;
}

/* The 1 024-thread workgroups of the streamed LDS walks come two to a CU = 8 waves per SIMD, and that is decided by SGPRs as
 * much as by VGPRs and LDS: a SIMD has 800, a wave is given its count rounded up to 16 plus 16 more the runtime reserves (trap
 * handler), so 8 waves fit only while the kernel needs <= 80.  At 82 the second workgroup no longer fits and the kernel runs at
 * HALF occupancy — which hipModuleOccupancyMaxActiveBlocksPerMultiprocessor does not report (it answers 2) and only the counters
 * show (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES 32 instead of 63).  Measured: the shadow walk 48.3 ms per four batches at 78 SGPRs, 61.2 at
 * 82 (profiles/r03_slp.txt).  So the compiler is held to 80 (it spills nothing: the excess was address arithmetic it can redo). */
#define RPT_LDS_WALK_SGPRS 80
/* LAST: the launch that traces the last extension ray of every path of a batch of known length WITHOUT NEE (the host's choice, rpt_hip.hip launch_iteration:
 * the same condition as the shade stage's last_iteration).  At that bounce the reference reads three things off the walk's result (kernels/src/lib.rs:62-109):
 * a miss adds the sky; a hit on the front of a triangle whose material emits adds its emission; any other hit adds nothing and ends the sample.  A ray that
 * passes the Moller-Trumbore test of NO emissive triangle (sc.last_emit_tri, at most RPT_LAST_EMIT_MAX of them, tested when the lane takes the ray) cannot
 * end on one, so "hit or miss" is all its walk has to say: it stops at its first accepted triangle (lds_walk_run MIXED) — in a closed scene about half of
 * the node visits of the bounce (tools/last_bounce_sim.py).  Its hit record names THAT triangle: not the nearest one, but like the nearest one not an
 * emitter, which is all the shade stage's last iteration looks at.  A ray that does pass such a test runs the reference's walk to its end. */
/* FIRST: the launch of a render call's first iteration, where every ray is a camera ray and leaves cfg.cam_position (k_path.h camera_ray; lib.rs:36-60):
 * the workgroup stages the plane records with that origin subtracted (lds_walk_run PRESUB) — twelve of the ~ 98 instructions of a node-pair step. */
#define RPT_NEAREST_PLAIN 0
#define RPT_NEAREST_LAST 1
#define RPT_NEAREST_FIRST 2
template <int STACK, int THREADS, int MODE = RPT_NEAREST_PLAIN>
__attribute__((amdgpu_num_sgpr(RPT_LDS_WALK_SGPRS)))
__global__ __launch_bounds__(THREADS) void k_traverse_nearest_stream(DevScene sc, DevState st, DevQueues q, uint32_t iteration,
                                                                       uint32_t SPAN /* slots a workgroup fetches at a time */,
                                                                       float cam_x, float cam_y, float cam_z /* FIRST: the origin of every ray of the launch */) {
    constexpr bool LAST = MODE == RPT_NEAREST_LAST, FIRST = MODE == RPT_NEAREST_FIRST;
    constexpr uint32_t NW = THREADS / RPT_WAVE;
    __shared__ uint16_t lds_stack[NW][STACK][RPT_WAVE];
    __shared__ WgPool pool;
    float4 *lds_scene = rpt_lds_dyn;
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    uint32_t *global_next = &q.count[Q_POOL0 + (iteration & 1u) * Q_LINE];
    if (blockIdx.x == 0u && threadIdx.x == 0u) {
        /* per-iteration bookkeeping, as in k_traverse_nearest (+ the other parity's slot counter, unused in this launch) */
        iteration_bookkeeping(q, iteration);
        q.count[Q_POOL0 + ((iteration + 1u) & 1u) * Q_LINE] = 0u;
    }
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    if (threadIdx.x == 0u) {
        const uint32_t g = atomicAdd(global_next, SPAN);
        pool.word = g < st.n_slots ? ((unsigned long long)(g + SPAN < st.n_slots ? g + SPAN : st.n_slots) << 32) | g : 0x00000000f0000000ull;
        pool.lock = 0u;
    }
    __syncthreads();
    if ((uint32_t)(pool.word >> 32) == 0u) return;             /* block-uniform: a late workgroup, nothing left */
    /* LAST: behind the image, the pair records (planes + child descriptors) of the flipped copy the hit-or-miss lanes walk in fixed order, when the
     * scene has one and the host found room for it (sc.last_flip_vecs float4; 0: those lanes walk the primary image near child first) */
    if (LAST)
        for (uint32_t k = threadIdx.x; k < sc.last_flip_vecs; k += THREADS) lds_scene[sc.lds_vecs + k] = sc.lds_image_last[k];
    if (FIRST) {
        const uint32_t P2 = 2u * sc.lds_pairs;                 /* image layout (rpt_hip.hip build_lds_image): 2 P plane records per axis, x | y | z */
        for (uint32_t k = threadIdx.x; k < 3u * P2; k += THREADS) {
            const float4 v = sc.lds_image[k];
            const float o = k < P2 ? cam_x : (k < 2u * P2 ? cam_y : cam_z);
            lds_scene[k] = make_float4(v.x - o, v.y - o, v.z - o, v.w - o);
        }
        for (uint32_t k = 3u * P2 + threadIdx.x; k < sc.lds_vecs; k += THREADS) lds_scene[k] = sc.lds_image[k];
        __syncthreads();
    }
    const SceneViewLds view = FIRST ? SceneViewLds{lds_scene, sc.lds_pairs, sc.n_triangles, sc.lds_root} : stage_scene_lds<THREADS>(sc, lds_scene);
    const float4 *img_lane = view.img;                         /* (per lane, LAST only) */
    uint32_t stop_first = 0u;
    float order_bias = 0.0f;
    uint16_t *stack = &lds_stack[wave][0][lane];
    F3 ro = f3(0, 0, 0), rd = f3(1, 1, 1), ird = f3(1, 1, 1);
    LdsWalk w;
    lds_walk_begin(view, w);
    w.cur = LDS_DESC_DEAD;
    uint32_t slot = 0u;
    bool have = false;                                         /* this lane holds a ray whose result is not written yet */
    bool pool_open = true;                                     /* wave-uniform: the launch may still have slots */
    uint32_t traced = 0u;                                      /* wave-uniform */
    /* LAST: a hit-or-miss lane that hit knows what the shade stage's last iteration would find out from the triangle's material — not an emitter, the sample
     * is finished with the radiance it has — and with several slots per pixel says so itself (HIT_DONE: k_shade.h, the `else` of the last iteration) */
    const bool done_here = LAST && st.group_shift != 0u;
    auto last_word = [&](const HitRecord &r, uint32_t stopped) {
        return (done_here && stopped != 0u && r.tri != HIT_MISS) ? make_float2(0.0f, __uint_as_float(HIT_DONE)) : make_float2(r.t, __uint_as_float(r.tri));
    };
    for (;;) {
        const unsigned long long idle_m = rpt_ballot(w.cur == LDS_DESC_DEAD);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if (pool_open && n_idle >= (uint32_t)RPT_STREAM_REFILL) {
            uint32_t base = 0u, got = 0u;
            bool finished = false;
            if (lane == 0u) base = wg_pool_take(&pool, global_next, st.n_slots, SPAN, n_idle, got, finished);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            pool_open = __builtin_amdgcn_readfirstlane((int)finished) == 0;
            bool took = false;
            if (w.cur == LDS_DESC_DEAD) {
                if (have) {
                    st.hit[slot] = last_word(w.res, stop_first);
                    have = false;
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                if (rank < got) {
                    const uint32_t cand = base + rank;
                    if (__float_as_uint(st.hit[cand].y) == HIT_PENDING) {
                        const float4 ra = st.ray_a[cand];
                        const float2 rb = st.ray_b[cand];
                        ro = f3(ra.x, ra.y, ra.z); rd = f3(ra.w, rb.x, rb.y);
                        slot = cand;
                        took = true;
                        if (fastdiv_ray_ok(sc.fastdiv_ok, ro, rd)) {
                            ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                            lds_walk_begin(view, w);
                            have = true;
                            if (LAST) {
                                bool may_emit = false;
                                for (uint32_t e = 0; e < sc.last_emit_n; ++e) {
                                    float t_e;
                                    bool bf_e;
                                    may_emit = moller_trumbore_view(view, sc.last_emit_tri[e], ro, rd, t_e, bf_e) || may_emit;
                                }
                                stop_first = may_emit ? 0u : 1u;
                                const bool flipped = !may_emit && sc.last_flip_vecs != 0u;
                                img_lane = flipped ? view.img + sc.lds_vecs : view.img;
                                order_bias = flipped ? __builtin_inff() : 0.0f;
                            }
                        } else {
                            /* outside the exact-division guard (a zero / denormal-small direction component): walked here, alone */
                            LdsWalk alone;
                            lds_walk_begin(view, alone);
                            lds_walk_run<STACK, false, false, false, false, FIRST>(view, alone, ro, rd, rd, 0.0f, stack, 0x7fffffff);
                            st.hit[cand] = make_float2(alone.res.t, __uint_as_float(alone.res.tri));
                        }
                    }
                }
            }
            traced += (uint32_t)__popcll(rpt_ballot(took));
            if (got != 0u || !pool_open) continue;             /* slots that were not pending leave lanes idle: look again */
            if (idle_m == ~0ull) { __builtin_amdgcn_s_sleep(8); continue; }   /* another wave is fetching the next span */
        }
        if (idle_m == ~0ull) {
            if (!pool_open) break;                             /* nothing in flight and nothing left to hand out */
            continue;
        }
        if (LAST) lds_walk_run<STACK, false, true, false, true>(view, w, ro, rd, ird, 0.0f, stack, pool_open ? RPT_STREAM_TRIPS : 0x7fffffff, img_lane, stop_first, order_bias);
        else lds_walk_run<STACK, false, true, false, false, FIRST>(view, w, ro, rd, ird, 0.0f, stack, pool_open ? RPT_STREAM_TRIPS : 0x7fffffff);
    }
    if (have) st.hit[slot] = last_word(w.res, stop_first);
    /* ray accounting + the alive flag, once per wave */
    if (lane == 0u && traced != 0u) {
        raise_flag(&q.count[Q_ALIVE0 + (iteration & 1u) * Q_LINE]);
        atomicAdd(&q.ray_shards[(blockIdx.x % RPT_STAT_SHARDS) * RPT_STAT_STRIDE], (unsigned long long)traced);
    }
}

/* Shadow rays (kernels/src/light_pick.rs:141-148): any-hit over the positions of the
 * shadow queue (k_common.h: sharded, dense up to the shards' tails); if unoccluded the pre-weighted NEE contribution is added to the
 * path's radiance (lib.rs:164).  A path that ended at this bounce (bit 31 of
 * the tag) is finished here: accumulated and, if samples remain, regenerated
 * in place (its slot becomes HIT_PENDING again). */
template <int STACK, bool LDS_SCENE, int THREADS, bool SMALL = false>
__global__ __launch_bounds__(THREADS) void k_traverse_shadow(DevScene sc, DevState st, DevQueues q, DevConfig cfg, DevStats *stats) {
    /* LDS-resident scenes walk 16-bit descriptors: 16-bit stack entries (32 KB per 1024-thread workgroup, which with a
     * <= 32 KB scene image is the 64 KB a workgroup may hold: 2 workgroups = 32 waves per CU) */
    typedef typename StackElem<LDS_SCENE || SMALL>::type StackT;
    __shared__ StackT lds_stack[THREADS / RPT_WAVE][STACK][RPT_WAVE];
    float4 *lds_scene = rpt_lds_dyn;
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    uint32_t i = blockIdx.x * THREADS + threadIdx.x;
    uint32_t positions, n;
    q_extent(q.shadow_cnt, positions, n);
    if (i == 0u && n) atomicAdd(&stats->shadow_rays, (unsigned long long)n);
    if (blockIdx.x * THREADS >= positions) return;             /* block-uniform */
    const auto view = stage_scene<LDS_SCENE, THREADS>(sc, lds_scene);
    if (i >= positions || !q_filled(q.shadow_cnt, i)) return;
    float4 o = q.sh_o[i], d = q.sh_d[i];
    uint32_t tag = __float_as_uint(d.w);
    uint32_t slot = tag & 0x7fffffffu;
    bool finish = (tag >> 31) != 0u;
    StackT *stack = &lds_stack[threadIdx.x / RPT_WAVE][0][threadIdx.x % RPT_WAVE];
    HitRecord h = traverse_one<STACK, true>(view, sc.fastdiv_ok, f3(o.x, o.y, o.z), f3(d.x, d.y, d.z), o.w, stack);
    bool visible = h.tri == HIT_MISS;
    if (visible || finish) {
        float4 r4 = st.rad[slot];
        F3 radiance = f3(r4.x, r4.y, r4.z);
        if (visible) {
            float4 c = q.sh_c[i];
            radiance = radiance + mask_nan3(f3(c.x, c.y, c.z));
        }
        if (finish) {
            finish_in_side_stage(st, cfg, slot, radiance, __float_as_uint(r4.w));
        } else {
            r4.x = radiance.x; r4.y = radiance.y; r4.z = radiance.z;
            st.rad[slot] = r4;
        }
    }
}

/* Shadow rays of an LDS-resident scene, streamed like the extension rays above (persistent workgroups, spans of the dense
 * shadow queue from a launch-wide counter, refill of finished lanes).  An any-hit walk cannot be pruned by max_t (the
 * reference prunes boxes against result.t = 1e6 until something is accepted, intersection.rs:212-213, and box-t / triangle-t
 * round differently), so an unoccluded ray crosses every box along its line while an occluded one may stop after two
 * visits: lane utilisation of the one-ray-per-lane kernel was 40 % (profiles/r02base_darkcornell_mis_pmc_sq.txt).
 * Lanes only record "occluded" in the unused .w of the entry's contribution record; k_shadow_resolve then adds the NEE
 * terms in one dense pass (all lanes busy, none of the walk's registers live). */
template <int STACK, int THREADS, bool FIXED /* fixed left-first order over the flipped image (shadow_order.h) */>
__attribute__((amdgpu_num_sgpr(RPT_LDS_WALK_SGPRS)))
__global__ __launch_bounds__(THREADS) void k_traverse_shadow_stream(DevScene sc, DevState st, DevQueues q, DevStats *stats, uint32_t SPAN) {
    constexpr uint32_t NW = THREADS / RPT_WAVE;
    __shared__ uint16_t lds_stack[NW][STACK][RPT_WAVE];
    __shared__ WgPool pool;
    float4 *lds_scene = rpt_lds_dyn;
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    uint32_t n, n_entries;                                     /* n: queue positions to hand out */
    q_extent(q.shadow_cnt, n, n_entries);
    uint32_t *global_next = &q.count[Q_SPOOL];                 /* zeroed by the shade stage of this iteration */
    if (blockIdx.x == 0u && threadIdx.x == 0u && n_entries) atomicAdd(&stats->shadow_rays, (unsigned long long)n_entries);
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    if (threadIdx.x == 0u) {
        const uint32_t g = n ? atomicAdd(global_next, SPAN) : 0u;
        pool.word = g < n ? ((unsigned long long)(g + SPAN < n ? g + SPAN : n) << 32) | g : 0x00000000f0000000ull;
        pool.lock = 0u;
    }
    __syncthreads();
    if ((uint32_t)(pool.word >> 32) == 0u) return;             /* block-uniform: nothing (left) to trace */
    const SceneViewLds view = stage_scene_lds<THREADS>(sc, lds_scene, FIXED);
    uint16_t *stack = &lds_stack[wave][0][lane];
    F3 ro = f3(0, 0, 0), rd = f3(1, 1, 1), ird = f3(1, 1, 1);
    float max_t = 0.0f;
    LdsWalk w;
    lds_walk_begin(view, w);
    w.cur = LDS_DESC_DEAD;
    uint32_t entry = 0u;
    bool have = false;
    bool pool_open = true;                                     /* wave-uniform */
    for (;;) {
        const unsigned long long idle_m = rpt_ballot(w.cur == LDS_DESC_DEAD);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if (pool_open && n_idle >= (uint32_t)RPT_STREAM_REFILL) {
            uint32_t base = 0u, got = 0u;
            bool finished = false;
            if (lane == 0u) base = wg_pool_take(&pool, global_next, n, SPAN, n_idle, got, finished);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            pool_open = __builtin_amdgcn_readfirstlane((int)finished) == 0;
            if (w.cur == LDS_DESC_DEAD) {
                if (have) {
                    q.sh_c[entry].w = w.res.tri == HIT_MISS ? 0.0f : 1.0f;
                    have = false;
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                if (rank < got && q_filled(q.shadow_cnt, base + rank)) {   /* (a position in the tail of a shard may be empty) */
                    entry = base + rank;
                    const float4 o = q.sh_o[entry], d = q.sh_d[entry];
                    ro = f3(o.x, o.y, o.z); rd = f3(d.x, d.y, d.z);
                    max_t = o.w;
                    have = true;
                    if (fastdiv_ray_ok(sc.fastdiv_ok, ro, rd)) {
                        ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        lds_walk_begin(view, w);
                    } else {
                        w.res = traverse_loop_lds<STACK, true, false, FIXED>(view, ro, rd, rd, max_t, stack);   /* alone; recorded at the next refill */
                    }
                }
            }
            if (got != 0u || !pool_open) continue;
            if (idle_m == ~0ull) { __builtin_amdgcn_s_sleep(8); continue; }   /* another wave is fetching the next span */
        }
        if (idle_m == ~0ull) {
            if (!pool_open) break;                             /* nothing in flight and nothing left to hand out */
            continue;
        }
        lds_walk_run<STACK, true, true, FIXED>(view, w, ro, rd, ird, max_t, stack, pool_open ? RPT_STREAM_TRIPS : 0x7fffffff);
    }
    if (have) q.sh_c[entry].w = w.res.tri == HIT_MISS ? 0.0f : 1.0f;
}

/* ---- streamed walks through GLOBAL memory (scenes too large for LDS) ----------------------------------------
 * Measured on MI355X (profiles/r02base_*): with one ray per lane the global-memory walk runs at 26 % (VeachMIS nearest),
 * 29 % (PBRTest) and 22 % (VeachMIS shadow) lane utilisation while two thirds of its wave cycles wait on L1/L2 — an
 * open scene leaves most slots of a wave without a pending ray after the first bounce (their paths ended in the sky),
 * and any-hit walks end after anything between one and a hundred node visits.  One-wave workgroups make the remedy
 * cheap: a wave owns SPAN consecutive slots (queue entries), and
 *   - (nearest) first compacts the pending ones into a wave-local LDS list — ballot + mbcnt, no atomic, the pool
 *     counter is a scalar register;
 *   - walks with a trip budget and, when RPT_GSTREAM_REFILL lanes are idle, lets them write their results and take
 *     the next rays of the list.
 * Per ray nothing changes (same tests, same order); slots stay identity mapped. */
#ifndef RPT_GSTREAM_RAYS
#define RPT_GSTREAM_RAYS 8         /* most slots per lane of a wave (the host lowers it for small launches) */
#endif
/* The nearest-hit walk streams better over a longer list — its pending list costs LDS (2 bytes per slot), and LDS is what caps
 * the waves of these kernels, so only where the stack is small: 16 slots per lane with a 16-bit stack of <= 24 entries (3 KB
 * + 2 KB per wave: still 8 waves per SIMD).  Measured, PBRTest traverse per 4 batches: 8 / 12 / 16 / 24 slots per lane
 * 92.9 / 89.0 / 86.9 / 95.1 ms; with a 32-entry stack 16 slots cost (the stand-in 439 -> 468 ms), and the any-hit walk
 * prefers 8 everywhere (VeachMIS shadow 56.6 / 58.0 / 57.6 / 60.8). */
#ifndef RPT_GSTREAM_RAYS_NEAREST_SMALL
#define RPT_GSTREAM_RAYS_NEAREST_SMALL 16
#endif
__host__ __device__ constexpr int gstream_rays_nearest(int stack, int width) {
    return (stack <= 24 && width <= 21) ? RPT_GSTREAM_RAYS_NEAREST_SMALL : RPT_GSTREAM_RAYS;
}
#ifndef RPT_GSTREAM_TRIPS
#define RPT_GSTREAM_TRIPS 8
#endif
/* (measured and dropped, round 3: dealing a span's rays grouped by the octant of their direction — the slots of a wave belong to
 * one or two pixels, so after a bounce their rays leave almost one point — 2 M-node stand-in + 2.4 %, PBRTest - 1.3 %, VeachMIS - 0.8 %,
 * the fat-leaf stand-in +- 0) */
#ifndef RPT_GSTREAM_REFILL_FIRST
#define RPT_GSTREAM_REFILL_FIRST 64      /* nearest-hit walk, iteration 0 of a batch: see k_traverse_nearest_gstream */
#endif
#ifndef RPT_GSTREAM_REFILL
#define RPT_GSTREAM_REFILL 24      /* (round 3, 64 pixels per wave: 8 / 16 / 24 idle lanes: PBRTest 7 390 / 7 390 / 7 445, VeachMIS 6 560 / 6 615 / 6 655 Mrays/s;
                                      trips 4 / 8 / 12 / 16: 7 355 / 7 390 / 7 320 / 7 250 and 6 620 / 6 615 / 6 530 / 6 480) */
#endif
/* The global-memory walks wait on memory two thirds of their cycles (profiles/r02_*_pmc_sq.txt) and live on occupancy.  Left
 * alone the compiler settles at 68 / 77 VGPRs (7 / 6 waves per SIMD); asked for 8 it needs 57 / 58 and spills nothing:
 * PBRTest traverse 97.3 -> 92.8 ms per 4 batches, VeachMIS traverse + shadow 91.8 -> 87.9, the 1 M-triangle stand-in's
 * shadow stage 391 -> 366.  (Wider stack entries cap the occupancy through LDS instead: hence the 24-bit form, WaveStack.) */
#ifndef RPT_GSTREAM_WAVES
#define RPT_GSTREAM_WAVES 8
#endif
#ifndef RPT_GSTREAM_WAVES_COOP
#define RPT_GSTREAM_WAVES_COOP 8   /* the fat-leaf build holds a leaf's triangle records in registers: 63 / 64 VGPRs, no spill.  Requesting the NEXT
                                      leaf's records one leaf ahead (9 more registers) was measured and lost at every occupancy: the 1 M-triangle
                                      stand-in 2 343 Mrays/s without, 2 008 / 2 164 / 2 099 with it at 8 (spilling) / 7 / 6 waves per SIMD */
#endif
__host__ __device__ constexpr int gstream_waves(int stack, int width, bool coop) {
    return (width <= 21 || (width == 24 && stack <= 24)) ? (coop ? (width >= 21 ? 7 : RPT_GSTREAM_WAVES_COOP) : RPT_GSTREAM_WAVES) : 1;   /* (where LDS allows it at all;
                                                             fat leaves + 21-bit entries: 8 waves would spill 18 registers, + 24-bit entries: 3) */
}
/* (8 waves per SIMD also need <= 80 SGPRs, see RPT_LDS_WALK_SGPRS: the builds the shipped scenes and the stand-ins use have 78; some of the
 * others — 21- and 32-bit stack entries — have 81 and run 7.  amdgpu_num_sgpr takes a literal, not a template expression, so it cannot follow
 * gstream_waves.) */
/* XCD-aware span mapping was measured on these kernels and rejected (profiles/r03_deepbvh_experiments.txt): workgroup id i runs on XCD
 * i % 8, so span = id spreads neighbouring pixels over all eight L2s.  Giving each XCD one contiguous eighth of the launch: 2 x
 * SLOWER on the 1 M-triangle stand-in (the XCD that owns the expensive part of the image finishes alone); runs of 64 consecutive
 * spans per XCD inside groups of 512: +-0; runs of 512: -14 %.  The identity mapping stays. */
template <int STACK, int WIDTH /* bits of a stack entry: 16, 21, 24, 32 */, bool COOP /* the scene has leaves of more than RPT_COOP_LEAF_MIN triangles */>
__attribute__((amdgpu_waves_per_eu(gstream_waves(STACK, WIDTH, COOP), 8)))
 __global__ __launch_bounds__(RPT_WAVE) void k_traverse_nearest_gstream(DevScene sc, DevState st, DevQueues q, uint32_t iteration,
                                                                       uint32_t SPAN /* slots per wave, <= 64 * gstream_rays_nearest(STACK, WIDTH) */) {
    __shared__ WaveStack<STACK, WIDTH> lds_stack;
    __shared__ uint16_t pend[RPT_WAVE * gstream_rays_nearest(STACK, WIDTH)];
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x == 0u && lane == 0u) {
        /* per-iteration bookkeeping, as in k_traverse_nearest */
        iteration_bookkeeping(q, iteration);
    }
    const uint32_t span_begin = blockIdx.x * SPAN;
    if (span_begin >= st.n_slots) return;
    const uint32_t span_end = span_begin + SPAN < st.n_slots ? span_begin + SPAN : st.n_slots;
    uint32_t count = 0u;                                       /* wave-uniform */
    /* the pending slots of the span, in slot order */
    for (uint32_t base = span_begin; base < span_end; base += RPT_WAVE) {
        const uint32_t s = base + lane;
        const bool p = s < span_end && __float_as_uint(st.hit[s].y) == HIT_PENDING;
        const unsigned long long m = rpt_ballot(p);
        if (p) pend[count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(s - span_begin);
        count += (uint32_t)__popcll(m);
    }
    if (count == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0u) {
        raise_flag(&q.count[Q_ALIVE0 + (iteration & 1u) * Q_LINE]);
        atomicAdd(&q.ray_shards[(blockIdx.x % RPT_STAT_SHARDS) * RPT_STAT_STRIDE], (unsigned long long)count);
    }
    typedef typename GstreamView<COOP>::type View;
    const View view = RPT_GSTREAM_VIEW(COOP, sc);
    auto stack = lds_stack.column(lane);
    F3 ro = f3(0, 0, 0), rd = f3(1, 1, 1), ird = f3(1, 1, 1);
    Walk<View> w;
    walk_begin(view, w);
    w.cur = View::dead();
    uint32_t slot = 0u, next = 0u;                             /* next: wave-uniform position in the list */
    bool have = false;                                         /* this lane holds a ray whose result is not written yet */
    /* The first iteration of a batch walks CAMERA rays: the 64 slots a wave deals together are one 8 x 8 pixel block at one sample index, their
     * rays stand on the same node step after step (k_traverse.h children_uniform: the scalar-cache path) and end within a few steps of each other.
     * A refill would put rays at the root beside rays deep in the tree and end that: there the wave takes its next 64 slots only when all are done. */
    const uint32_t refill_at = (View::kUniformScalar && iteration == 0u) ? (uint32_t)RPT_GSTREAM_REFILL_FIRST : (uint32_t)RPT_GSTREAM_REFILL;
    for (;;) {
        const unsigned long long idle_m = rpt_ballot(walk_dead(w));
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        const bool more = next < count;                        /* wave-uniform */
        if ((more && n_idle >= refill_at) || idle_m == ~0ull) {
            if (walk_dead(w)) {
                if (have) {
                    st.hit[slot] = make_float2(w.res.t, __uint_as_float(w.res.tri));
                    have = false;
                }
                const uint32_t at = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                if (at < count) {
                    slot = span_begin + pend[at];
                    const float4 ra = st.ray_a[slot];
                    const float2 rb = st.ray_b[slot];
                    ro = f3(ra.x, ra.y, ra.z); rd = f3(ra.w, rb.x, rb.y);
                    have = true;
                    if (fastdiv_ray_ok(sc.fastdiv_ok, ro, rd)) {
                        ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        walk_begin(view, w);
                    } else {
                        /* outside the exact-division guard (a zero / denormal-small direction component): walked here, alone;
                         * the lane stays idle and writes the result at its next refill */
                        w.res = traverse_loop<STACK, false, false>(view, ro, rd, rd, 0.0f, stack);
                    }
                }
            }
            if (!more) break;                                  /* everything handed out, walked and written */
            next += n_idle;
            continue;
        }
        walk_run<STACK, false, true>(view, w, ro, rd, ird, 0.0f, stack, more ? RPT_GSTREAM_TRIPS : 0x7fffffff);
    }
}

/* what k_traverse_shadow does with the result of one shadow ray (light_pick.rs:148 + lib.rs:164) */
__device__ __forceinline__ void shadow_resolve(const DevState &st, const DevQueues &q, const DevConfig &cfg, uint32_t entry, uint32_t tag,
                                               bool visible) {
    const uint32_t slot = tag & 0x7fffffffu;
    const bool finish = (tag >> 31) != 0u;
    if (visible || finish) {
        float4 r4 = st.rad[slot];
        F3 radiance = f3(r4.x, r4.y, r4.z);
        if (visible) {
            float4 c = q.sh_c[entry];
            radiance = radiance + mask_nan3(f3(c.x, c.y, c.z));
        }
        if (finish) {
            finish_in_side_stage(st, cfg, slot, radiance, __float_as_uint(r4.w));
        } else {
            r4.x = radiance.x; r4.y = radiance.y; r4.z = radiance.z;
            st.rad[slot] = r4;
        }
    }
}

/* second half of the streamed LDS shadow stage: one dense pass over the shadow queue */
__global__ __launch_bounds__(RPT_BLOCK) void k_shadow_resolve(DevState st, DevQueues q, DevConfig cfg) {
    if (q.count[Q_DRAINED] != 0u) return;
    const uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    uint32_t positions, n;
    q_extent(q.shadow_cnt, positions, n);
    if (i >= positions || !q_filled(q.shadow_cnt, i)) return;
    shadow_resolve(st, q, cfg, i, __float_as_uint(q.sh_d[i].w), q.sh_c[i].w == 0.0f);
}

/* Shadow rays, streamed: the queue is dense already (up to the tails of its shards); a wave owns SPAN consecutive positions and refills lanes whose
 * any-hit walk has ended (found an occluder after two visits, or crossed the whole scene without one).  Lanes only note
 * "occluded" per entry in LDS while walking; the NEE terms are added afterwards in one dense pass over the span (all
 * lanes busy, and the registers of the walk are dead by then: 61 instead of 91 VGPRs). */
template <int STACK, int WIDTH, bool COOP, bool FIXED /* fixed left-first order over the flipped pair array (shadow_order.h) */>
__attribute__((amdgpu_waves_per_eu(gstream_waves(STACK, WIDTH, COOP), 8)))
 __global__ __launch_bounds__(RPT_WAVE) void k_traverse_shadow_gstream(DevScene sc, DevState st, DevQueues q, DevConfig cfg, DevStats *stats,
                                                                      uint32_t SPAN) {
    __shared__ WaveStack<STACK, WIDTH> lds_stack;
    __shared__ uint8_t occluded[RPT_WAVE * RPT_GSTREAM_RAYS];
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    const uint32_t lane = threadIdx.x;
    uint32_t n, n_entries;                                     /* n: queue positions of the launch */
    q_extent(q.shadow_cnt, n, n_entries);
    if (blockIdx.x == 0u && lane == 0u && n_entries) atomicAdd(&stats->shadow_rays, (unsigned long long)n_entries);
    const uint32_t begin = blockIdx.x * SPAN;
    if (begin >= n) return;
    const uint32_t end = begin + SPAN < n ? begin + SPAN : n;
    {
        typedef typename GstreamView<COOP>::type View;
        const View view = FIXED ? RPT_GSTREAM_VIEW_SHADOW(COOP, sc) : RPT_GSTREAM_VIEW(COOP, sc);
        auto stack = lds_stack.column(lane);
        F3 ro = f3(0, 0, 0), rd = f3(1, 1, 1), ird = f3(1, 1, 1);
        float max_t = 0.0f;
        Walk<View> w;
        walk_begin(view, w);
        w.cur = View::dead();
        uint32_t entry = 0u, next = begin;                     /* next: wave-uniform */
        bool have = false;
        for (;;) {
            const unsigned long long idle_m = rpt_ballot(walk_dead(w));
            const uint32_t n_idle = (uint32_t)__popcll(idle_m);
            const bool more = next < end;                      /* wave-uniform */
            if ((more && n_idle >= (uint32_t)RPT_GSTREAM_REFILL) || idle_m == ~0ull) {
                if (walk_dead(w)) {
                    if (have) {
                        occluded[entry - begin] = w.res.tri == HIT_MISS ? (uint8_t)0 : (uint8_t)1;
                        have = false;
                    }
                    const uint32_t at = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                    if (at < end && q_filled(q.shadow_cnt, at)) {
                        const float4 o = q.sh_o[at], d = q.sh_d[at];
                        ro = f3(o.x, o.y, o.z); rd = f3(d.x, d.y, d.z);
                        max_t = o.w;
                        entry = at;
                        have = true;
                        if (fastdiv_ray_ok(sc.fastdiv_ok, ro, rd)) {
                            ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                            walk_begin(view, w);
                        } else {
                            w.res = traverse_loop<STACK, true, false, FIXED>(view, ro, rd, rd, max_t, stack);   /* alone; noted at the next refill */
                        }
                    }
                }
                if (!more) break;
                next += n_idle;
                continue;
            }
            walk_run<STACK, true, true, FIXED>(view, w, ro, rd, ird, max_t, stack, more ? RPT_GSTREAM_TRIPS : 0x7fffffff);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = begin; base < end; base += RPT_WAVE) {
        const uint32_t e = base + lane;
        if (e < end && q_filled(q.shadow_cnt, e)) shadow_resolve(st, q, cfg, e, __float_as_uint(q.sh_d[e].w), occluded[e - begin] == 0u);
    }
}

/* Test hook: plain ray arrays in, hit arrays out (rpt_debug_trace_rays). */
template <int STACK, bool ANY_HIT, bool LDS_SCENE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_trace_debug(DevScene sc, uint32_t n, const float *origins, const float *dirs,
                                                         const float *max_t, float *out_t, uint32_t *out_tri, uint32_t *out_flags) {
    /* LDS-resident scenes walk 16-bit descriptors: 16-bit stack entries (32 KB per 1024-thread workgroup, which with a
     * <= 32 KB scene image is the 64 KB a workgroup may hold: 2 workgroups = 32 waves per CU) */
    typedef typename StackElem<LDS_SCENE>::type StackT;
    __shared__ StackT lds_stack[THREADS / RPT_WAVE][STACK][RPT_WAVE];
    float4 *lds_scene = rpt_lds_dyn;
    uint32_t i = blockIdx.x * THREADS + threadIdx.x;
    const auto view = stage_scene<LDS_SCENE, THREADS>(sc, lds_scene);
    if (i >= n) return;
    F3 ro = f3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
    F3 rd = f3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    StackT *stack = &lds_stack[threadIdx.x / RPT_WAVE][0][threadIdx.x % RPT_WAVE];
    HitRecord h = traverse_one<STACK, ANY_HIT>(view, sc.fastdiv_ok, ro, rd, ANY_HIT ? max_t[i] : 0.0f, stack);
    out_t[i] = h.t;
    out_tri[i] = (h.tri == HIT_MISS) ? 0u : (h.tri & 0x7fffffffu);
    out_flags[i] = (h.tri == HIT_MISS) ? 0u : (1u | ((h.tri >> 31) << 1));
}

#endif /* RPT_K_TRAVERSE_H */
