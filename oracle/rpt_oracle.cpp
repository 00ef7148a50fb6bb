/*
 * rpt_oracle.cpp — CPU restatement of the reference's path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (librpt_hip.so,
 * librpt_host.so, the Python package) links, loads or calls this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
 * as the checker / reported baseline.
 *
 * It follows the reference one function at a time, in the reference's own
 * operation order (one megakernel-style trace_pixel per pixel-sample, exactly
 * like the reference's CPU path), and deliberately shares NO code with the
 * wavefront HIP kernels except rpt_math.h (deterministic transcendentals, which
 * must be common for bit parity) — so that HIP-vs-oracle parity is evidence.
 *
 * Reference lines restated (all under /root/reference):
 *   kernels/src/lib.rs:21-186          trace_pixel
 *   kernels/src/intersection.rs:9-54   muller_trumbore
 *   kernels/src/intersection.rs:104-122 intersect_aabb
 *   kernels/src/intersection.rs:169-234 intersect_nearest / intersect_any
 *   kernels/src/intersection.rs:77-101 brute force (test hook only)
 *   kernels/src/bsdf.rs:179-387        PBR, get_pbr_bsdf
 *   kernels/src/light_pick.rs:8-199    NEE + MIS
 *   kernels/src/rng.rs:20-63           additive-recurrence LDS
 *   kernels/src/skybox.rs:8-94         procedural sky
 *   kernels/src/util.rs (live subset)  sampling / microfacet helpers
 *   shared_structs/src/image_polyfill.rs:32-55  CPU image sampler
 *   src/trace.rs:273-308               trace_cpu sample loop (row parallel)
 * Third-party arithmetic restated from its published definition (not vendored
 * in the reference): glam 0.22.0 Vec2/Vec3/Vec4/Mat3 operator order
 * (Cargo.lock:1008-1015); Rust float->int `as` casts; f32::{min,max,clamp,powi}.
 *
 * PARITY PINNING: the reference cannot be compiled here (no Rust toolchain), so
 * bit-level parity with the real reference is UNPINNED.  This oracle is pinned
 * by the reference's own known answers: the furnace test
 * (tests/correctness_tests.rs:14-33, both NEE modes), the struct layouts, the
 * LDS/blue-noise integer KATs of SURVEY.md Appendix B.3, and BVH == brute force.
 *
 * Math backend: rpt_math.h by default (bit-identical to the GPU); compile with
 * -DORACLE_USE_LIBM to call the platform libm exactly as the Rust CPU path
 * would, to measure how far the shared math is from "what glibc gives".
 *
 * Documented deviations from the reference:
 *   - alias-table index clamped to len-1 when gen_r1() == 1.0 (the reference
 *     panics on the CPU); occurrences are counted.
 *   - traversal stack overflow (> 32) sets an error flag instead of panicking.
 */
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rpt/shared_structs.h"
#include "../rust-path-tracer_amd/csrc/rpt_math.h"

namespace {

/* ------------------------------------------------------------------------ */
/* math backend                                                              */
/* ------------------------------------------------------------------------ */
#ifdef ORACLE_USE_LIBM
inline float m_sin(float x) { return sinf(x); }
inline float m_cos(float x) { return cosf(x); }
inline float m_acos(float x) { return acosf(x); }
inline float m_asin(float x) { return asinf(x); }
inline float m_atan2(float y, float x) { return atan2f(y, x); }
inline float m_atan(float x) { return atanf(x); }
inline float m_exp(float x) { return expf(x); }
inline float m_exp_sky(float x) { return expf(x); }
inline float m_pow(float x, float y) { return powf(x, y); }
#else
inline float m_sin(float x) { return rptm::sinr(x); }
inline float m_cos(float x) { return rptm::cosr(x); }
inline float m_acos(float x) { return rptm::acosr(x); }
inline float m_asin(float x) { return rptm::asinr(x); }
inline float m_atan2(float y, float x) { return rptm::atan2r(y, x); }
inline float m_atan(float x) { return rptm::atanr(x); }
inline float m_exp(float x) { return rptm::expr(x); }
inline float m_exp_sky(float x) { return rptm::exp_sky(x); }     /* skybox.rs only: see rpt_math.h */
inline float m_pow(float x, float y) { return rptm::powr(x, y); }
#endif
inline float m_sqrt(float x) { return __builtin_sqrtf(x); }
inline float m_min(float a, float b) { return rptm::fminr(a, b); }   /* f32::min */
inline float m_max(float a, float b) { return rptm::fmaxr(a, b); }   /* f32::max */
inline float m_clamp(float x, float lo, float hi) {                  /* f32::clamp */
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

constexpr float PI_F = 3.14159265358979323846f;   /* core::f32::consts::PI */
constexpr float EPS = 0.001f;                     /* util.rs:5 */

/* ------------------------------------------------------------------------ */
/* glam 0.22 scalar vector types, operator order as published                 */
/* ------------------------------------------------------------------------ */
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 splat3(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(V3 a) { return m_sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }          /* v * length_recip() */
inline V3 lerp3(V3 a, V3 b, float s) { return a + ((b - a) * s); }   /* Vec3::lerp */
inline bool is_finite3(V3 a) { return rptm::finiter(a.x) && rptm::finiter(a.y) && rptm::finiter(a.z); }
inline bool ne_zero3(V3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }
inline float max_element(V3 a) { return m_max(a.x, m_max(a.y, a.z)); }
inline V3 xyz(const float *p) { return V3{p[0], p[1], p[2]}; }

inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, V2 b) { return V2{a.x * b.x, a.y * b.y}; }
inline V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
inline V2 operator*(V2 a, float s) { return V2{a.x * s, a.y * s}; }

inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 lerp4(V4 a, V4 b, float s) { return a + ((b - a) * s); }

struct M3 { V3 c0, c1, c2; };   /* column major */
inline V3 mul(const M3 &m, V3 v) {          /* Mat3::mul_vec3 */
    V3 r = m.c0 * v.x;
    r = r + (m.c1 * v.y);
    r = r + (m.c2 * v.z);
    return r;
}
inline M3 mul(const M3 &a, const M3 &b) { return M3{mul(a, b.c0), mul(a, b.c1), mul(a, b.c2)}; }
inline M3 rotation_y(float angle) {
    float s = m_sin(angle), c = m_cos(angle);
    return M3{v3(c, 0.0f, -s), v3(0.0f, 1.0f, 0.0f), v3(s, 0.0f, c)};
}
inline M3 rotation_x(float angle) {
    float s = m_sin(angle), c = m_cos(angle);
    return M3{v3(1.0f, 0.0f, 0.0f), v3(0.0f, c, s), v3(0.0f, -s, c)};
}

/* ------------------------------------------------------------------------ */
/* kernels/src/rng.rs:20-63                                                   */
/* ------------------------------------------------------------------------ */
const uint32_t LDS_PRIMES[32] = {
    0x6a09e667u, 0xbb67ae84u, 0x3c6ef372u, 0xa54ff539u, 0x510e527fu, 0x9b05688au, 0x1f83d9abu, 0x5be0cd18u,
    0xcbbb9d5cu, 0x629a2929u, 0x91590159u, 0x452fecd8u, 0x67332667u, 0x8eb44a86u, 0xdb0c2e0bu, 0x47b5481du,
    0xae5f9155u, 0xcf6c85d1u, 0x2f73477du, 0x6d1826cau, 0x8b43d455u, 0xe360b595u, 0x1c456002u, 0x6f196330u,
    0xd94ebeafu, 0x9cc4a611u, 0x261dc1f2u, 0x5815a7bdu, 0x70b7ed67u, 0xa1513c68u, 0x44f93634u, 0x720dcdfcu};

inline float lds(uint32_t n, uint32_t dimension, uint32_t offset) {
    const float INV_U32_MAX_FLOAT = 1.0f / 4294967296.0f;
    uint32_t v = LDS_PRIMES[dimension & 31u] * (n + offset);   /* wrapping_mul(wrapping_add) */
    return (float)v * INV_U32_MAX_FLOAT;
}

struct RngState {
    uint32_t n, offset;
    uint32_t dimension;
    bool overflow;
    float gen_r1() {
        dimension += 1;
        if (dimension > 31) { overflow = true; }    /* reference: index panic */
        return lds(n, dimension, offset);
    }
    V2 gen_r2() { float a = gen_r1(); float b = gen_r1(); return V2{a, b}; }
    V3 gen_r3() { float a = gen_r1(); float b = gen_r1(); float c = gen_r1(); return V3{a, b, c}; }
};

/* ------------------------------------------------------------------------ */
/* scene view + counters                                                      */
/* ------------------------------------------------------------------------ */
struct Image {
    const float *texels;   /* Vec4 per texel */
    uint32_t width, height;
};

struct Scene {
    const rpt_per_vertex_data *per_vertex; size_t n_vertices;
    const rpt_triangle *indices; size_t n_triangles;
    const rpt_bvh_node *nodes; size_t n_nodes;
    const rpt_material_data *materials; size_t n_materials;
    const rpt_light_pick_entry *light_pick; size_t n_light_pick;
    Image atlas, skybox;
};

struct Counters {
    uint64_t extension_rays = 0, shadow_rays = 0, sky_evals = 0, light_index_clamped = 0;
    uint64_t node_pops = 0, box_tests = 0, tri_tests = 0;
    uint32_t max_stack = 0;
    uint32_t error_flags = 0;   /* bit0 stack overflow, bit1 rng dimension overflow */
    uint8_t *event_log = nullptr;   /* analysis hook, see log_event */
    uint32_t event_cap = 0, event_len = 0;
    uint64_t *hist_nearest = nullptr, *hist_any = nullptr;   /* analysis hook: node pops per node (oracle_node_histogram) */
    uint32_t *node_log = nullptr;                            /* analysis hook: the popped node indices of one walk (oracle_trace_nodes) */
    uint32_t node_cap = 0, node_len = 0;
};

/* ------------------------------------------------------------------------ */
/* shared_structs/src/image_polyfill.rs:32-55                                 */
/* ------------------------------------------------------------------------ */
inline V4 sample_raw(const Image &img, int32_t cx, int32_t cy) {
    /* `coord.x as usize % width as usize`: i32 -> usize sign-extends */
    uint64_t x = (uint64_t)(int64_t)cx % (uint64_t)img.width;
    uint64_t y = (uint64_t)(int64_t)cy % (uint64_t)img.height;
    const float *t = img.texels + 4 * (y * (uint64_t)img.width + x);
    return V4{t[0], t[1], t[2], t[3]};
}
inline V4 sample_by_lod(const Image &img, V2 coord) {
    V2 scaled = coord * V2{(float)img.width, (float)img.height};
    V2 fl = V2{rptm::floorr(scaled.x), rptm::floorr(scaled.y)};
    V2 frac = scaled - fl;                                   /* Vec2::fract = v - v.floor() */
    int32_t cx = rptm::f2i32_sat(rptm::ceilr(scaled.x)), cy = rptm::f2i32_sat(rptm::ceilr(scaled.y));
    int32_t fx = rptm::f2i32_sat(fl.x), fy = rptm::f2i32_sat(fl.y);
    V4 c00 = sample_raw(img, fx, fy);
    V4 c01 = sample_raw(img, fx, cy);
    V4 c10 = sample_raw(img, cx, fy);
    V4 c11 = sample_raw(img, cx, cy);
    V4 a = lerp4(c00, c10, frac.x);
    V4 b = lerp4(c01, c11, frac.x);
    return lerp4(a, b, frac.y);
}

/* ------------------------------------------------------------------------ */
/* kernels/src/intersection.rs                                                */
/* ------------------------------------------------------------------------ */
inline bool muller_trumbore(V3 ro, V3 rd, V3 a, V3 b, V3 c, float &out_t, bool &out_backface) {
    out_t = 0.0f;
    V3 edge1 = b - a;
    V3 edge2 = c - a;
    V3 pv = cross(rd, edge2);
    float det = dot(edge1, pv);
    out_backface = (rptm::f2u(det) >> 31) != 0;   /* num_traits Signed::is_negative == sign bit */
    if (rptm::absr(det) < 1e-6f) return false;
    float inv_det = 1.0f / det;
    V3 tv = ro - a;
    float u = dot(tv, pv) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qv = cross(tv, edge1);
    float v = dot(rd, qv) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(edge2, qv) * inv_det;
    if (t < 0.0f) return false;
    out_t = t;
    return true;
}

struct TraceResult {
    rpt_triangle triangle{0, 0, 0, 0};
    uint32_t triangle_index = 0;
    float t = 1000000.0f;
    bool hit = false;
    bool backface = false;
};

inline float intersect_aabb(V3 bmin, V3 bmax, V3 ro, V3 rd, float prev_min_t) {
    float tx1 = (bmin.x - ro.x) / rd.x;
    float tx2 = (bmax.x - ro.x) / rd.x;
    float tmin = m_min(tx1, tx2);
    float tmax = m_max(tx1, tx2);
    float ty1 = (bmin.y - ro.y) / rd.y;
    float ty2 = (bmax.y - ro.y) / rd.y;
    tmin = m_max(tmin, m_min(ty1, ty2));
    tmax = m_min(tmax, m_max(ty1, ty2));
    float tz1 = (bmin.z - ro.z) / rd.z;
    float tz2 = (bmax.z - ro.z) / rd.z;
    tmin = m_max(tmin, m_min(tz1, tz2));
    tmax = m_min(tmax, m_max(tz1, tz2));
    if (tmax >= tmin && tmax > 0.0f && tmin < prev_min_t) return tmin;
    return INFINITY;
}

/* optional event log for tools/traversal_sim.py: one byte per node visit, 0 = inner node, 1 = leaf */
inline void log_event(Counters &cnt, uint8_t e) {
    if (!cnt.event_log) return;
    if (cnt.event_len < cnt.event_cap) cnt.event_log[cnt.event_len] = e;
    cnt.event_len += 1;
}

template <bool NEAREST_HIT>
TraceResult intersect_front_to_back(const Scene &sc, V3 ro, V3 rd, float max_t, Counters &cnt) {
    uint32_t stack[32];
    uint32_t len = 0;
    stack[len++] = 0;
    TraceResult result;
    while (len != 0) {
        uint32_t node_index = stack[--len];
        const rpt_bvh_node &node = sc.nodes[node_index];
        cnt.node_pops++;
        log_event(cnt, node.triangle_count > 0 ? 1 : 0);
        if (uint64_t *h = NEAREST_HIT ? cnt.hist_nearest : cnt.hist_any) h[node_index] += 1;
        if (cnt.node_log) { if (cnt.node_len < cnt.node_cap) cnt.node_log[cnt.node_len] = node_index; cnt.node_len += 1; }
        if (node.triangle_count > 0) {
            for (uint32_t i = 0; i < node.triangle_count; ++i) {
                uint32_t triangle_index = node.left_or_first + i;
                rpt_triangle tri = sc.indices[triangle_index];
                V3 a = xyz(sc.per_vertex[tri.v0].vertex);
                V3 b = xyz(sc.per_vertex[tri.v1].vertex);
                V3 c = xyz(sc.per_vertex[tri.v2].vertex);
                float t = 0.0f;
                bool backface = false;
                cnt.tri_tests++;
                if (muller_trumbore(ro, rd, a, b, c, t, backface) && t > 0.001f && t < result.t &&
                    (NEAREST_HIT || t <= max_t)) {
                    result.triangle = tri;
                    result.triangle_index = triangle_index;
                    result.t = m_min(result.t, t);
                    result.hit = true;
                    result.backface = backface;
                    if (!NEAREST_HIT) return result;
                }
            }
        } else {
            uint32_t min_index = node.left_or_first;
            uint32_t max_index = node.left_or_first + 1;
            const rpt_bvh_node &min_child = sc.nodes[min_index];
            const rpt_bvh_node &max_child = sc.nodes[max_index];
            float min_dist = intersect_aabb(xyz(min_child.aabb_min), xyz(min_child.aabb_max), ro, rd, result.t);
            float max_dist = intersect_aabb(xyz(max_child.aabb_min), xyz(max_child.aabb_max), ro, rd, result.t);
            cnt.box_tests += 2;
            if (min_dist > max_dist) {
                uint32_t ti = min_index; min_index = max_index; max_index = ti;
                float td = min_dist; min_dist = max_dist; max_dist = td;
            }
            if (rptm::isinfr(min_dist)) continue;
            if (rptm::finiter(max_dist)) {
                if (len >= 32) { cnt.error_flags |= 1u; return result; }
                stack[len++] = max_index;
            }
            if (len >= 32) { cnt.error_flags |= 1u; return result; }
            stack[len++] = min_index;
            if (len > cnt.max_stack) cnt.max_stack = len;
        }
    }
    return result;
}

/* intersection.rs:77-101, used only by the BVH == brute-force test hook */
TraceResult intersect_brute_force(const Scene &sc, V3 ro, V3 rd) {
    TraceResult result;
    for (size_t i = 0; i < sc.n_triangles; ++i) {
        rpt_triangle tri = sc.indices[i];
        V3 a = xyz(sc.per_vertex[tri.v0].vertex);
        V3 b = xyz(sc.per_vertex[tri.v1].vertex);
        V3 c = xyz(sc.per_vertex[tri.v2].vertex);
        float t = 0.0f;
        bool backface = false;
        if (muller_trumbore(ro, rd, a, b, c, t, backface) && t > 0.001f && t < result.t) {
            result.triangle = tri;
            result.triangle_index = (uint32_t)i;
            result.t = m_min(result.t, t);
            result.hit = true;
            result.backface = backface;
        }
    }
    return result;
}

/* ------------------------------------------------------------------------ */
/* kernels/src/util.rs (live subset)                                          */
/* ------------------------------------------------------------------------ */
inline V3 cosine_sample_hemisphere(float r1, float r2) {
    float theta = m_acos(m_sqrt(r1));
    float phi = 2.0f * PI_F * r2;
    return v3(m_sin(theta) * m_cos(phi), m_cos(theta), m_sin(theta) * m_sin(phi));
}
inline void create_cartesian(V3 up, V3 &o_up, V3 &o_right, V3 &o_forward) {
    V3 arbitrary = v3(0.1f, 0.5f, 0.9f);
    V3 temp_vec = normalize(cross(up, arbitrary));
    V3 right = normalize(cross(temp_vec, up));
    V3 forward = normalize(cross(up, right));
    o_up = up; o_right = right; o_forward = forward;
}
inline V3 reflect(V3 i, V3 normal) { return i - normal * 2.0f * dot(i, normal); }
inline float ggx_distribution(V3 normal, V3 halfway, float roughness) {
    float numerator = roughness * roughness;
    float n_dot_h = m_max(dot(normal, halfway), 0.0f);
    float denominator = (n_dot_h * n_dot_h) * (numerator - 1.0f) + 1.0f;
    denominator = m_max(PI_F * (denominator * denominator), EPS);
    return numerator / denominator;
}
inline V3 sample_ggx(float r1, float r2, V3 reflection_direction, float roughness) {
    float a = roughness * roughness;
    float phi = 2.0f * PI_F * r1;
    float cos_theta = m_sqrt((1.0f - r2) / (r2 * (a * a - 1.0f) + 1.0f));
    float sin_theta = m_sqrt(1.0f - cos_theta * cos_theta);
    V3 halfway = v3(m_cos(phi) * sin_theta, m_sin(phi) * sin_theta, cos_theta);
    V3 up = rptm::absr(reflection_direction.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    V3 tangent = normalize(cross(up, reflection_direction));
    V3 bitangent = cross(reflection_direction, tangent);
    return normalize(tangent * halfway.x + bitangent * halfway.y + reflection_direction * halfway.z);
}
inline float geometry_schlick_ggx(V3 normal, V3 view_direction, float roughness) {
    float numerator = m_max(dot(normal, view_direction), 0.0f);
    float r = (roughness * roughness) / 8.0f;
    float denominator = numerator * (1.0f - r) + r;
    return numerator / denominator;
}
inline float geometry_smith_schlick_ggx(V3 normal, V3 view_direction, V3 light_direction, float roughness) {
    return geometry_schlick_ggx(normal, view_direction, roughness) *
           geometry_schlick_ggx(normal, light_direction, roughness);
}
inline V3 fresnel_schlick(float cos_theta, V3 f0) {
    return f0 + (splat3(1.0f) - f0) * rptm::powi5(1.0f - cos_theta);
}
inline float fresnel_schlick_scalar(float in_ior, float out_ior, float cos_theta) {
    float f0 = rptm::powi2((in_ior - out_ior) / (in_ior + out_ior));
    return f0 + (1.0f - f0) * rptm::powi5(1.0f - cos_theta);
}
inline V3 barycentric(V3 p, V3 a, V3 b, V3 c) {
    V3 v0 = b - a, v1 = c - a, v2 = p - a;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1);
    float d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    return v3(1.0f - v - w, v, w);
}
inline float power_heuristic(float p1, float p2) {
    float p1_2 = p1 * p1;
    return p1_2 / (p1_2 + p2 * p2);
}
inline V3 mask_nan(V3 v) { return is_finite3(v) ? v : splat3(0.0f); }
inline float lerp_f(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* ------------------------------------------------------------------------ */
/* kernels/src/bsdf.rs:11-26, 179-387                                         */
/* ------------------------------------------------------------------------ */
enum Lobe : uint32_t { DiffuseReflection = 0, SpecularReflection = 1 };

struct BSDFSample {
    float pdf = 0.0f;
    Lobe sampled_lobe = DiffuseReflection;
    V3 spectrum{0, 0, 0};
    V3 sampled_direction{0, 0, 0};
};

const float DIELECTRIC_IOR = 1.5f;
const float DIELECTRIC_F0_SQRT = (DIELECTRIC_IOR - 1.0f) / (DIELECTRIC_IOR + 1.0f);
const float DIELECTRIC_F0 = DIELECTRIC_F0_SQRT * DIELECTRIC_F0_SQRT;

struct PBR {
    V3 albedo;
    float roughness, metallic;
    V2 specular_weight_clamp;

    V3 evaluate_diffuse_fast(float cos_theta, float specular_weight, V3 ks) const {
        V3 kd = (splat3(1.0f) - ks) * (1.0f - metallic);
        V3 diffuse = kd * albedo / PI_F;
        return diffuse * cos_theta / (1.0f - specular_weight);
    }
    V3 evaluate_specular_fast(V3 view_direction, V3 normal, V3 sample_direction, float cos_theta, float d_term,
                              float specular_weight, V3 ks) const {
        float g_term = geometry_smith_schlick_ggx(normal, view_direction, sample_direction, roughness);
        V3 specular_numerator = d_term * g_term * ks;
        float specular_denominator = 4.0f * m_max(dot(normal, view_direction), 0.0f) * cos_theta;
        V3 specular = specular_numerator / m_max(specular_denominator, EPS);
        return specular * cos_theta / specular_weight;
    }
    float pdf_diffuse_fast(float cos_theta) const { return cos_theta / PI_F; }
    float pdf_specular_fast(V3 view_direction, V3 normal, V3 halfway, float d_term) const {
        return (d_term * dot(normal, halfway)) / (4.0f * dot(view_direction, halfway));
    }
    float specular_weight_of(V3 view_direction, V3 normal) const {
        float approx_fresnel = fresnel_schlick_scalar(1.0f, DIELECTRIC_IOR, m_max(dot(normal, view_direction), 0.0f));
        float specular_weight = lerp_f(approx_fresnel, 1.0f, metallic);
        if (specular_weight != 0.0f && specular_weight != 1.0f)
            specular_weight = m_clamp(specular_weight, specular_weight_clamp.x, specular_weight_clamp.y);
        return specular_weight;
    }

    V3 evaluate(V3 view_direction, V3 normal, V3 sample_direction, Lobe lobe_type) const {
        float specular_weight = specular_weight_of(view_direction, normal);
        float cos_theta = m_max(dot(normal, sample_direction), 0.0f);
        V3 halfway = normalize(view_direction + sample_direction);
        V3 f0 = lerp3(splat3(DIELECTRIC_F0), albedo, metallic);
        V3 ks = fresnel_schlick(m_max(dot(halfway, view_direction), 0.0f), f0);
        if (lobe_type == DiffuseReflection) return evaluate_diffuse_fast(cos_theta, specular_weight, ks);
        float d_term = ggx_distribution(normal, halfway, roughness);
        return evaluate_specular_fast(view_direction, normal, sample_direction, cos_theta, d_term, specular_weight, ks);
    }

    BSDFSample sample(V3 view_direction, V3 normal, RngState &rng) const {
        V3 rng_sample = rng.gen_r3();
        float specular_weight = specular_weight_of(view_direction, normal);
        V3 sampled_direction;
        Lobe sampled_lobe;
        if (rng_sample.z >= specular_weight) {
            V3 up, nt, nb;
            create_cartesian(normal, up, nt, nb);
            V3 s = cosine_sample_hemisphere(rng_sample.x, rng_sample.y);
            sampled_direction = normalize(v3(s.x * nb.x + s.y * up.x + s.z * nt.x,
                                             s.x * nb.y + s.y * up.y + s.z * nt.y,
                                             s.x * nb.z + s.y * up.z + s.z * nt.z));
            sampled_lobe = DiffuseReflection;
        } else {
            V3 reflection_direction = reflect(-view_direction, normal);
            sampled_direction = sample_ggx(rng_sample.x, rng_sample.y, reflection_direction, roughness);
            sampled_lobe = SpecularReflection;
        }
        float cos_theta = m_max(dot(normal, sampled_direction), EPS);
        V3 halfway = normalize(view_direction + sampled_direction);
        V3 f0 = lerp3(splat3(DIELECTRIC_F0), albedo, metallic);
        V3 ks = fresnel_schlick(m_max(dot(halfway, view_direction), 0.0f), f0);
        BSDFSample out;
        out.sampled_direction = sampled_direction;
        out.sampled_lobe = sampled_lobe;
        if (sampled_lobe == DiffuseReflection) {
            out.pdf = pdf_diffuse_fast(cos_theta);
            out.spectrum = evaluate_diffuse_fast(cos_theta, specular_weight, ks);
        } else {
            float d_term = ggx_distribution(normal, halfway, roughness);
            out.pdf = pdf_specular_fast(view_direction, normal, halfway, d_term);
            out.spectrum = evaluate_specular_fast(view_direction, normal, sampled_direction, cos_theta, d_term,
                                                  specular_weight, ks);
        }
        return out;
    }

    float pdf(V3 view_direction, V3 normal, V3 sample_direction, Lobe lobe_type) const {
        if (lobe_type == DiffuseReflection) {
            float cos_theta = m_max(dot(normal, sample_direction), 0.0f);
            return pdf_diffuse_fast(cos_theta);
        }
        V3 halfway = normalize(view_direction + sample_direction);
        float d_term = ggx_distribution(normal, halfway, roughness);
        return pdf_specular_fast(view_direction, normal, halfway, d_term);
    }
};

inline PBR get_pbr_bsdf(const rpt_tracing_config &config, const rpt_material_data &material, V2 uv, const Image &atlas) {
    PBR bsdf;
    if (material.has_albedo_texture != 0) {
        V2 scaled_uv = V2{material.albedo[0], material.albedo[1]} + uv * V2{material.albedo[2], material.albedo[3]};
        V4 a = sample_by_lod(atlas, scaled_uv);
        bsdf.albedo = v3(a.x, a.y, a.z);
    } else {
        bsdf.albedo = xyz(material.albedo);
    }
    float roughness, metallic;
    if (material.has_roughness_texture != 0) {
        V2 scaled_uv = V2{material.roughness[0], material.roughness[1]} + uv * V2{material.roughness[2], material.roughness[3]};
        roughness = sample_by_lod(atlas, scaled_uv).x;
    } else {
        roughness = material.roughness[0];
    }
    if (material.has_metallic_texture != 0) {
        V2 scaled_uv = V2{material.metallic[0], material.metallic[1]} + uv * V2{material.metallic[2], material.metallic[3]};
        metallic = sample_by_lod(atlas, scaled_uv).x;
    } else {
        metallic = material.metallic[0];
    }
    bsdf.roughness = m_max(roughness, EPS);
    bsdf.metallic = m_min(metallic, 1.0f - EPS);
    bsdf.specular_weight_clamp = V2{config.specular_weight_clamp[0], config.specular_weight_clamp[1]};
    return bsdf;
}

/* ------------------------------------------------------------------------ */
/* kernels/src/light_pick.rs                                                  */
/* ------------------------------------------------------------------------ */
struct DirectLightSample {
    float light_area = 0.0f;
    V3 light_normal{0, 0, 0};
    float light_pick_pdf = 0.0f;
    V3 light_emission{0, 0, 0};
    uint32_t light_triangle_index = 0;
    V3 throughput{0, 0, 0};
    V3 direct_light_contribution{0, 0, 0};
};

inline float calculate_light_pdf(float light_area, float light_distance, V3 light_normal, V3 light_direction) {
    float cos_theta = dot(light_normal, -light_direction);
    if (cos_theta <= 0.0f) return 0.0f;
    return rptm::powi2(light_distance) / (light_area * cos_theta);
}
inline float get_weight(uint32_t nee_mode, float p1, float p2) {
    return nee_mode == RPT_NEE_MIS ? power_heuristic(p1, p2) : 1.0f;
}

extern thread_local float *g_shadow_dump;
extern thread_local uint32_t g_cur_bounce, g_ray_dump_bounce;
extern thread_local bool g_shadow_dump_hit;
extern thread_local uint64_t *g_dead_shadow_rays;
DirectLightSample sample_direct_lighting(uint32_t nee_mode, const Scene &sc, V3 throughput, const PBR &surface_bsdf,
                                         V3 surface_point, V3 surface_normal, V3 ray_direction, RngState &rng,
                                         Counters &cnt) {
    DirectLightSample info;
    if (sc.light_pick[0].ratio < 0.0f) return info;   /* sentinel */

    /* pick_light (light_pick.rs:8-16) */
    V2 r = rng.gen_r2();
    uint32_t len = (uint32_t)sc.n_light_pick;
    uint32_t idx = rptm::f2u32_sat(r.x * (float)len);
    if (idx >= len) { idx = len - 1; cnt.light_index_clamped++; }   /* Appendix C deviation */
    const rpt_light_pick_entry &entry = sc.light_pick[idx];
    uint32_t light_index;
    float light_area, light_pick_pdf;
    if (r.y < entry.ratio) {
        light_index = entry.triangle_index_a; light_area = entry.triangle_area_a; light_pick_pdf = entry.triangle_pick_pdf_a;
    } else {
        light_index = entry.triangle_index_b; light_area = entry.triangle_area_b; light_pick_pdf = entry.triangle_pick_pdf_b;
    }
    rpt_triangle lt = sc.indices[light_index];
    V3 va = xyz(sc.per_vertex[lt.v0].vertex), vb = xyz(sc.per_vertex[lt.v1].vertex), vc = xyz(sc.per_vertex[lt.v2].vertex);
    V3 na = xyz(sc.per_vertex[lt.v0].normal), nb = xyz(sc.per_vertex[lt.v1].normal), nc = xyz(sc.per_vertex[lt.v2].normal);
    V3 light_normal = (na + nb + nc) / 3.0f;
    V3 light_emission = xyz(sc.materials[lt.material].emissive);

    /* pick_triangle_point (light_pick.rs:19-23) */
    V2 r2 = rng.gen_r2();
    float r1_sqrt = m_sqrt(r2.x);
    V3 light_point = (1.0f - r1_sqrt) * va + (r1_sqrt * (1.0f - r2.y)) * vb + (r1_sqrt * r2.y) * vc;
    V3 light_direction_unorm = light_point - surface_point;
    float light_distance = length(light_direction_unorm);
    V3 light_direction = light_direction_unorm / light_distance;

    V3 direct = splat3(0.0f);
    cnt.shadow_rays++;
    if (g_shadow_dump && g_cur_bounce == g_ray_dump_bounce) {
        V3 so = surface_point + light_direction * EPS;
        g_shadow_dump[0] = so.x; g_shadow_dump[1] = so.y; g_shadow_dump[2] = so.z;
        g_shadow_dump[3] = light_direction.x; g_shadow_dump[4] = light_direction.y; g_shadow_dump[5] = light_direction.z;
        g_shadow_dump[6] = light_distance - EPS * 2.0f; g_shadow_dump[7] = (float)idx;
        g_shadow_dump_hit = true;
    }
    TraceResult light_trace = intersect_front_to_back<false>(sc, surface_point + light_direction * EPS, light_direction,
                                                             light_distance - EPS * 2.0f, cnt);
    if (!light_trace.hit) {
        float light_pdf = calculate_light_pdf(light_area, light_distance, light_normal, light_direction);
        if (light_pdf > 0.0f) {
            V3 bsdf_attenuation = surface_bsdf.evaluate(-ray_direction, surface_normal, light_direction, DiffuseReflection);
            float bsdf_pdf = surface_bsdf.pdf(-ray_direction, surface_normal, light_direction, DiffuseReflection);
            if (bsdf_pdf > 0.0f) {
                float weight = get_weight(nee_mode, light_pdf, bsdf_pdf);
                direct = (bsdf_attenuation * light_emission * weight / light_pdf) / light_pick_pdf;
            }
        }
    }
    if (g_dead_shadow_rays) {
        /* analysis hook (tools/dead_shadow_rays.py): would this shadow ray's term be zero whatever the walk finds?  (the term an UNOCCLUDED ray adds:
         * light_pdf <= 0 — the light faces away — or bsdf_pdf <= 0 — the light is below the surface's horizon — or a masked non-finite product) */
        V3 would = splat3(0.0f);
        float light_pdf = calculate_light_pdf(light_area, light_distance, light_normal, light_direction);
        if (light_pdf > 0.0f) {
            V3 att = surface_bsdf.evaluate(-ray_direction, surface_normal, light_direction, DiffuseReflection);
            float bsdf_pdf = surface_bsdf.pdf(-ray_direction, surface_normal, light_direction, DiffuseReflection);
            if (bsdf_pdf > 0.0f) would = (att * light_emission * get_weight(nee_mode, light_pdf, bsdf_pdf) / light_pdf) / light_pick_pdf;
        }
        V3 term = mask_nan(throughput * would);
        g_dead_shadow_rays[0] += 1;
        if (term.x == 0.0f && term.y == 0.0f && term.z == 0.0f) {
            g_dead_shadow_rays[1] += 1;
            if (light_trace.hit) g_dead_shadow_rays[2] += 1;
            if (g_shadow_dump && g_cur_bounce == g_ray_dump_bounce) g_shadow_dump[7] = -1.0f;      /* (dump: marked as deciding nothing) */
        }
        if (light_trace.hit) g_dead_shadow_rays[3] += 1;
    }
    info.light_area = light_area;
    info.light_normal = light_normal;
    info.light_pick_pdf = light_pick_pdf;
    info.light_emission = light_emission;
    info.light_triangle_index = light_index;
    info.throughput = throughput;
    info.direct_light_contribution = throughput * direct;
    return info;
}

inline V3 calculate_bsdf_mis_contribution(const TraceResult &trace_result, const BSDFSample &last_bsdf_sample,
                                          const DirectLightSample &last_light_sample) {
    if (trace_result.triangle_index != last_light_sample.light_triangle_index) return splat3(0.0f);
    float light_pdf = calculate_light_pdf(last_light_sample.light_area, trace_result.t, last_light_sample.light_normal,
                                          last_bsdf_sample.sampled_direction);
    if (light_pdf > 0.0f) {
        float weight = power_heuristic(last_bsdf_sample.pdf, light_pdf);
        V3 direct = (last_bsdf_sample.spectrum * last_light_sample.light_emission * weight / last_bsdf_sample.pdf) /
                    last_light_sample.light_pick_pdf;
        return last_light_sample.throughput * direct;
    }
    return splat3(0.0f);
}

/* ------------------------------------------------------------------------ */
/* kernels/src/skybox.rs                                                      */
/* ------------------------------------------------------------------------ */
const V3 RAY_SCATTER_COEFF = {58e-7f, 135e-7f, 331e-7f};
const V3 RAY_EFFECTIVE_COEFF = RAY_SCATTER_COEFF;
const V3 MIE_SCATTER_COEFF = {2e-5f, 2e-5f, 2e-5f};
const V3 MIE_EFFECTIVE_COEFF = {2e-5f * 1.1f, 2e-5f * 1.1f, 2e-5f * 1.1f};
const float EARTH_RADIUS = 6360e3f;
const float ATMOSPHERE_RADIUS = 6380e3f;
const float H_RAY = 8e3f;
const float H_MIE = 12e2f;
const V3 CENTER = {0.0f, -EARTH_RADIUS, 0.0f};

inline float escape(V3 p, V3 d, float r) {
    V3 v = p - CENTER;
    float b = dot(v, d);
    float det = b * b - dot(v, v) + r * r;
    if (det < 0.0f) return -1.0f;
    det = m_sqrt(det);
    float t1 = -b - det;
    float t2 = -b + det;
    if (t1 >= 0.0f) return t1;
    return t2;
}
inline V2 densities_rm(V3 p) {
    float h = m_max(length(p - CENTER) - EARTH_RADIUS, 0.0f);
    return V2{m_exp_sky(-h / H_RAY), m_exp_sky(-h / H_MIE)};
}
inline V2 scatter_depth_int(V3 o, V3 d, float l) {
    return densities_rm(o) * (l / 2.0f) + densities_rm(o + d * l) * (l / 2.0f);
}
inline void scatter_in(V3 origin, V3 direction, float depth, uint32_t steps, V3 sundir, V3 &o_ir, V3 &o_im) {
    depth = depth / (float)steps;
    V3 i_r = splat3(0.0f), i_m = splat3(0.0f);
    V2 total_depth_rm = V2{0.0f, 0.0f};
    for (uint32_t i = 0; i < steps; ++i) {
        V3 p = origin + direction * (depth * (float)i);
        V2 d_rm = densities_rm(p) * depth;
        total_depth_rm = total_depth_rm + d_rm;
        V2 depth_rm_sum = total_depth_rm + scatter_depth_int(p, sundir, escape(p, sundir, ATMOSPHERE_RADIUS));
        V3 e = (-RAY_EFFECTIVE_COEFF) * depth_rm_sum.x - MIE_EFFECTIVE_COEFF * depth_rm_sum.y;
        V3 a = v3(m_exp_sky(e.x), m_exp_sky(e.y), m_exp_sky(e.z));
        i_r = i_r + a * d_rm.x;
        i_m = i_m + a * d_rm.y;
    }
    o_ir = i_r; o_im = i_m;
}
inline V3 sky_scatter(const float *sundir4, V3 origin, V3 direction) {
    V3 sundir = xyz(sundir4);
    V3 i_r, i_m;
    scatter_in(origin, direction, escape(origin, direction, ATMOSPHERE_RADIUS), 12, sundir, i_r, i_m);
    float mu = dot(direction, sundir);
    V3 res = (sundir4[3] * (1.0f + mu * mu)) *
             (i_r * RAY_EFFECTIVE_COEFF * 0.0597f + i_m * MIE_SCATTER_COEFF * 0.0196f / m_pow(1.58f - 1.52f * mu, 1.5f));
    V3 g = mask_nan(v3(m_sqrt(res.x), m_sqrt(res.y), m_sqrt(res.z)));
    return v3(m_pow(g.x, 2.2f), m_pow(g.y, 2.2f), m_pow(g.z, 2.2f));
}

/* ------------------------------------------------------------------------ */
/* kernels/src/lib.rs:21-186                                                  */
/* ------------------------------------------------------------------------ */
struct PixelResult { V4 radiance; uint32_t next_n, next_offset; };

/* optional dump of the extension ray of one chosen bounce (analysis hook) */
thread_local float *g_ray_dump = nullptr;      /* 6 floats: origin, direction */
thread_local uint32_t g_ray_dump_bounce = 0;
thread_local bool g_ray_dump_hit = false;
thread_local float *g_shadow_dump = nullptr;   /* analysis hook: 8 floats: origin, direction, max_t, light-table index */
thread_local uint32_t g_cur_bounce = 0;
thread_local bool g_shadow_dump_hit = false;
thread_local uint64_t *g_dead_shadow_rays = nullptr;   /* analysis hook: [shadow rays, of them with a zero term, of those occluded, all occluded] */
/* analysis hook (tools/shade_bin_sim.py): what the shade stage does with the path at each bounce — 1 miss, 2 ends on an emitter (lib.rs:86-109),
 * 3 diffuse lobe sampled, 4 specular lobe sampled; 0 = the path did not reach that bounce */
thread_local uint8_t *g_kind_log = nullptr;
inline void log_kind(uint32_t bounce, uint8_t kind) { if (g_kind_log && bounce < 8) g_kind_log[bounce] = kind; }

PixelResult trace_pixel(uint32_t id_x, uint32_t id_y, const rpt_tracing_config &config, rpt_rng_state rng,
                        const Scene &sc, Counters &cnt) {
    uint32_t nee_mode = config.nee <= 2 ? config.nee : 0;   /* NextEventEstimation::from_u32 */
    bool nee = nee_mode != RPT_NEE_NONE;
    RngState rng_state{rng.n, rng.offset, 0, false};

    V2 jitter = rng_state.gen_r2();
    V2 suv = V2{(float)id_x, (float)id_y} + jitter;
    V2 uv = V2{suv.x / (float)config.width, 1.0f - suv.y / (float)config.height} * 2.0f - V2{1.0f, 1.0f};
    uv.y *= (float)config.height / (float)config.width;

    V3 ray_origin = xyz(config.cam_position);
    V3 ray_direction = normalize(v3(uv.x, uv.y, 1.0f));
    M3 euler_mat = mul(rotation_y(config.cam_rotation[1]), rotation_x(config.cam_rotation[0]));
    ray_direction = mul(euler_mat, ray_direction);

    V3 throughput = splat3(1.0f);
    V3 radiance = splat3(0.0f);
    BSDFSample last_bsdf_sample;
    DirectLightSample last_light_sample;

    for (uint32_t bounce = 0; bounce < config.max_bounces; ++bounce) {
        cnt.extension_rays++;
        g_cur_bounce = bounce;
        if (g_ray_dump && bounce == g_ray_dump_bounce) {
            g_ray_dump[0] = ray_origin.x; g_ray_dump[1] = ray_origin.y; g_ray_dump[2] = ray_origin.z;
            g_ray_dump[3] = ray_direction.x; g_ray_dump[4] = ray_direction.y; g_ray_dump[5] = ray_direction.z;
            g_ray_dump_hit = true;
        }
        TraceResult trace_result = intersect_front_to_back<true>(sc, ray_origin, ray_direction, 0.0f, cnt);
        V3 hit = ray_origin + ray_direction * trace_result.t;

        if (!trace_result.hit) {
            log_kind(bounce, 1);
            cnt.sky_evals++;
            if (config.has_skybox == 0) {
                radiance = radiance + throughput * sky_scatter(config.sun_direction, ray_origin, ray_direction);
            } else {
                float rotation = m_atan2(config.sun_direction[2], config.sun_direction[0]);
                V3 rotated = mul(rotation_y(rotation), ray_direction);
                float u = 0.5f + m_atan2(rotated.z, rotated.x) / (2.0f * PI_F);
                float v = 1.0f - (0.5f + m_asin(rotated.y) / PI_F);
                float intensity = config.sun_direction[3] * (1.0f / 15.0f);
                V4 s = sample_by_lod(sc.skybox, V2{u, v});
                radiance = radiance + throughput * v3(s.x, s.y, s.z) * intensity;
            }
            break;
        }

        const rpt_material_data &material = sc.materials[trace_result.triangle.material];
        if (ne_zero3(xyz(material.emissive))) {
            log_kind(bounce, 2);
            if (trace_result.backface) break;
            if (!nee || bounce == 0 || last_bsdf_sample.sampled_lobe != DiffuseReflection) {
                radiance = radiance + mask_nan(throughput * xyz(material.emissive));
                break;
            }
            if (nee_mode == RPT_NEE_MIS && last_bsdf_sample.sampled_lobe == DiffuseReflection) {
                V3 direct_contribution = calculate_bsdf_mis_contribution(trace_result, last_bsdf_sample, last_light_sample);
                radiance = radiance + mask_nan(direct_contribution);
                break;
            }
        }

        const rpt_per_vertex_data &vda = sc.per_vertex[trace_result.triangle.v0];
        const rpt_per_vertex_data &vdb = sc.per_vertex[trace_result.triangle.v1];
        const rpt_per_vertex_data &vdc = sc.per_vertex[trace_result.triangle.v2];
        V3 bary = barycentric(hit, xyz(vda.vertex), xyz(vdb.vertex), xyz(vdc.vertex));
        V3 normal = bary.x * xyz(vda.normal) + bary.y * xyz(vdb.normal) + bary.z * xyz(vdc.normal);
        V2 tuv = bary.x * V2{vda.uv0[0], vda.uv0[1]} + bary.y * V2{vdb.uv0[0], vdb.uv0[1]} + bary.z * V2{vdc.uv0[0], vdc.uv0[1]};
        {
            /* uv.clamp(0,1) != uv  ->  uv.fract()  (lib.rs:127-129); Vec2::clamp = max(min).min(max) */
            float cx = m_min(m_max(tuv.x, 0.0f), 1.0f), cy = m_min(m_max(tuv.y, 0.0f), 1.0f);
            if (cx != tuv.x || cy != tuv.y) tuv = V2{tuv.x - rptm::floorr(tuv.x), tuv.y - rptm::floorr(tuv.y)};
        }

        if (material.has_normal_texture != 0) {
            V2 scaled_uv = V2{material.normals[0], material.normals[1]} + tuv * V2{material.normals[2], material.normals[3]};
            V4 nm4 = sample_by_lod(sc.atlas, scaled_uv) * 2.0f - V4{1.0f, 1.0f, 1.0f, 1.0f};
            V3 tangent = bary.x * xyz(vda.tangent) + bary.y * xyz(vdb.tangent) + bary.z * xyz(vdc.tangent);
            M3 tbn = M3{tangent, cross(tangent, normal), normal};
            normal = normalize(mul(tbn, v3(nm4.x, nm4.y, nm4.z)));
        }

        PBR bsdf = get_pbr_bsdf(config, material, tuv, sc.atlas);
        BSDFSample bsdf_sample = bsdf.sample(-ray_direction, normal, rng_state);
        last_bsdf_sample = bsdf_sample;
        log_kind(bounce, bsdf_sample.sampled_lobe == DiffuseReflection ? 3 : 4);

        if (nee && bsdf_sample.sampled_lobe == DiffuseReflection) {
            last_light_sample = sample_direct_lighting(nee_mode, sc, throughput, bsdf, hit, normal, ray_direction,
                                                       rng_state, cnt);
            radiance = radiance + mask_nan(last_light_sample.direct_light_contribution);
        }

        throughput = throughput * (bsdf_sample.spectrum / bsdf_sample.pdf);
        ray_direction = bsdf_sample.sampled_direction;
        ray_origin = hit + ray_direction * EPS;

        if (bounce > config.min_bounces) {
            float prob = max_element(throughput);
            if (rng_state.gen_r1() > prob) break;
            throughput = throughput * (1.0f / prob);
        }
    }
    if (rng_state.overflow) cnt.error_flags |= 2u;
    PixelResult out;
    out.radiance = V4{radiance.x, radiance.y, radiance.z, 1.0f};
    out.next_n = rng.n + 1;
    out.next_offset = rng.offset;
    return out;
}

}  // namespace

/* ======================================================================== */
/* C interface for tests / bench cpu_baseline                                 */
/* ======================================================================== */
extern "C" {

typedef struct oracle_scene {
    const rpt_per_vertex_data *per_vertex; size_t n_vertices;
    const rpt_triangle *indices; size_t n_triangles;
    const rpt_bvh_node *nodes; size_t n_nodes;
    const rpt_material_data *materials; size_t n_materials;
    const rpt_light_pick_entry *light_pick; size_t n_light_pick;
    const float *atlas_rgba32f; uint32_t atlas_w, atlas_h;     /* Vec4 texels, may be NULL */
    const float *skybox_rgba32f; uint32_t sky_w, sky_h;        /* Vec4 texels, may be NULL */
} oracle_scene;

typedef struct oracle_stats {
    uint64_t samples, extension_rays, shadow_rays, sky_evals, light_index_clamped;
    uint64_t node_pops, box_tests, tri_tests;
    uint32_t max_stack, error_flags;
    double seconds;
    uint32_t threads;
} oracle_stats;

static Scene make_scene(const oracle_scene *s) {
    static const float magenta[16] = {1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1};   /* asset.rs:283-290 */
    Scene sc;
    sc.per_vertex = s->per_vertex; sc.n_vertices = s->n_vertices;
    sc.indices = s->indices; sc.n_triangles = s->n_triangles;
    sc.nodes = s->nodes; sc.n_nodes = s->n_nodes;
    sc.materials = s->materials; sc.n_materials = s->n_materials;
    sc.light_pick = s->light_pick; sc.n_light_pick = s->n_light_pick;
    sc.atlas = s->atlas_rgba32f ? Image{s->atlas_rgba32f, s->atlas_w, s->atlas_h} : Image{magenta, 2, 2};
    sc.skybox = s->skybox_rgba32f ? Image{s->skybox_rgba32f, s->sky_w, s->sky_h} : Image{magenta, 2, 2};
    return sc;
}

const char *oracle_math_backend(void) {
#ifdef ORACLE_USE_LIBM
    return "libm";
#else
    return "rpt_math";
#endif
}

/* src/trace.rs:273-308: n_samples passes over all pixels (or the rectangle
 * [x0,x1) x [y0,y1) when restricted), rows in parallel; accum += (rgb,1);
 * rng.n += 1.  accum is width*height float4, rng is width*height. */
int oracle_trace_cpu(const rpt_tracing_config *config, const oracle_scene *scene, rpt_rng_state *rng, float *accum,
                     uint32_t n_samples, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, int n_threads,
                     oracle_stats *stats) {
    if (!config || !scene || !rng || !accum) return -1;
    Scene sc = make_scene(scene);
    const uint32_t W = config->width, H = config->height;
    if (x1 > W) x1 = W;
    if (y1 > H) y1 = H;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    std::vector<Counters> counters((size_t)n_threads);
    auto t_begin = std::chrono::steady_clock::now();
    for (uint32_t s = 0; s < n_samples; ++s) {
        std::atomic<uint32_t> next_row{y0};
        auto worker = [&](int tid) {
            Counters &cnt = counters[(size_t)tid];
            for (;;) {
                uint32_t y = next_row.fetch_add(1);
                if (y >= y1) break;
                for (uint32_t x = x0; x < x1; ++x) {
                    size_t i = (size_t)y * W + x;
                    PixelResult r = trace_pixel(x, y, *config, rng[i], sc, cnt);
                    accum[4 * i + 0] += r.radiance.x;
                    accum[4 * i + 1] += r.radiance.y;
                    accum[4 * i + 2] += r.radiance.z;
                    accum[4 * i + 3] += r.radiance.w;
                    rng[i].n = r.next_n;
                    rng[i].offset = r.next_offset;
                }
            }
        };
        if (n_threads == 1) {
            worker(0);
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
            for (auto &t : pool) t.join();
        }
    }
    auto t_end = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (const Counters &c : counters) {
            stats->extension_rays += c.extension_rays;
            stats->shadow_rays += c.shadow_rays;
            stats->sky_evals += c.sky_evals;
            stats->light_index_clamped += c.light_index_clamped;
            stats->node_pops += c.node_pops;
            stats->box_tests += c.box_tests;
            stats->tri_tests += c.tri_tests;
            if (c.max_stack > stats->max_stack) stats->max_stack = c.max_stack;
            stats->error_flags |= c.error_flags;
        }
        stats->samples = (uint64_t)(x1 - x0) * (y1 - y0) * n_samples;
        stats->seconds = std::chrono::duration<double>(t_end - t_begin).count();
        stats->threads = (uint32_t)n_threads;
    }
    return 0;
}

/* One pixel-sample, for fine-grained tests: returns radiance rgb. */
int oracle_trace_pixel(const rpt_tracing_config *config, const oracle_scene *scene, uint32_t x, uint32_t y,
                       rpt_rng_state rng, float *out_rgb) {
    Scene sc = make_scene(scene);
    Counters cnt;
    PixelResult r = trace_pixel(x, y, *config, rng, sc, cnt);
    out_rgb[0] = r.radiance.x; out_rgb[1] = r.radiance.y; out_rgb[2] = r.radiance.z;
    return (int)cnt.error_flags;
}

/* Ray batch through the BVH. mode 0 nearest, 1 any-hit(max_t), 2 brute force. */
int oracle_trace_rays(const oracle_scene *scene, int mode, size_t n, const float *origins, const float *dirs,
                      const float *max_t, float *out_t, uint32_t *out_tri, uint32_t *out_flags) {
    Scene sc = make_scene(scene);
    Counters cnt;
    for (size_t i = 0; i < n; ++i) {
        V3 ro = xyz(origins + 3 * i), rd = xyz(dirs + 3 * i);
        TraceResult r;
        if (mode == 0) r = intersect_front_to_back<true>(sc, ro, rd, 0.0f, cnt);
        else if (mode == 1) r = intersect_front_to_back<false>(sc, ro, rd, max_t[i], cnt);
        else r = intersect_brute_force(sc, ro, rd);
        out_t[i] = r.t;
        out_tri[i] = r.triangle_index;
        out_flags[i] = (r.hit ? 1u : 0u) | (r.backface ? 2u : 0u);
    }
    return (int)cnt.error_flags;
}

/* Analysis hook: per-ray sequence of node visits (0 inner / 1 leaf) of the reference traversal, and the rays
 * (origin, direction) each bounce of each pixel-sample produces, for tools/traversal_sim.py. */
int oracle_trace_events(const oracle_scene *scene, size_t n, const float *origins, const float *dirs, uint8_t *events,
                        uint32_t max_events, uint32_t *lengths) {
    Scene sc = make_scene(scene);
    Counters cnt;
    for (size_t i = 0; i < n; ++i) {
        cnt.event_log = events + i * (size_t)max_events;
        cnt.event_cap = max_events;
        cnt.event_len = 0;
        intersect_front_to_back<true>(sc, xyz(origins + 3 * i), xyz(dirs + 3 * i), 0.0f, cnt);
        lengths[i] = cnt.event_len;
    }
    return 0;
}

/* Analysis hook (tools/uniform_visit_share.py): per ray the sequence of popped node indices of the nearest-hit walk (max_t null) or of the
 * any-hit walk with the given max_t. */
int oracle_trace_nodes(const oracle_scene *scene, size_t n, const float *origins, const float *dirs, const float *max_t, uint32_t *nodes_out,
                       uint32_t max_nodes, uint32_t *lengths) {
    Scene sc = make_scene(scene);
    Counters cnt;
    for (size_t i = 0; i < n; ++i) {
        cnt.node_log = nodes_out + i * (size_t)max_nodes;
        cnt.node_cap = max_nodes;
        cnt.node_len = 0;
        if (max_t) intersect_front_to_back<false>(sc, xyz(origins + 3 * i), xyz(dirs + 3 * i), max_t[i], cnt);
        else intersect_front_to_back<true>(sc, xyz(origins + 3 * i), xyz(dirs + 3 * i), 0.0f, cnt);
        lengths[i] = cnt.node_len;
    }
    return 0;
}

/* shadow[(y*W + x)*8..] = (origin, direction, max_t, light-table index) of the shadow ray that bounce `bounce` of sample rng[i] of every pixel traces */
int oracle_dump_shadow_rays(const rpt_tracing_config *config, const oracle_scene *scene, const rpt_rng_state *rng, uint32_t bounce,
                            float *shadow, uint8_t *valid) {
    Scene sc = make_scene(scene);
    Counters cnt;
    g_ray_dump_bounce = bounce;
    for (uint32_t y = 0; y < config->height; ++y)
        for (uint32_t x = 0; x < config->width; ++x) {
            size_t i = (size_t)y * config->width + x;
            g_shadow_dump = shadow + 8 * i;
            g_shadow_dump_hit = false;
            trace_pixel(x, y, *config, rng[i], sc, cnt);
            valid[i] = g_shadow_dump_hit ? 1 : 0;
        }
    g_shadow_dump = nullptr;
    return 0;
}

/* Analysis hook (tools/dead_shadow_rays.py): out[4] = shadow rays of n_samples samples of the pixels of rect, those whose NEE term is zero whatever the walk
 * finds, those of them that are occluded, all occluded ones */
int oracle_dead_shadow_rays(const rpt_tracing_config *config, const oracle_scene *scene, const rpt_rng_state *rng, uint32_t n_samples, uint32_t stride, uint64_t *out) {
    Scene sc = make_scene(scene);
    Counters cnt;
    out[0] = out[1] = out[2] = out[3] = 0;
    g_dead_shadow_rays = out;
    for (uint32_t y = 0; y < config->height; y += stride)
        for (uint32_t x = 0; x < config->width; x += stride) {
            rpt_rng_state r = rng[(size_t)y * config->width + x];
            for (uint32_t s = 0; s < n_samples; ++s) { trace_pixel(x, y, *config, r, sc, cnt); r.n += 1; }
        }
    g_dead_shadow_rays = nullptr;
    return 0;
}

/* Analysis hook (tools/shade_bin_sim.py): kinds[i * 8 + b] = what bounce b of sample rng[i].n + sample of pixel pixels[i] (x | y << 16) is to the shade stage */
int oracle_path_kinds(const rpt_tracing_config *config, const oracle_scene *scene, const rpt_rng_state *rng_full, uint32_t sample,
                      const uint32_t *pixels, size_t n_pixels, uint8_t *kinds) {
    Scene sc = make_scene(scene);
    Counters cnt;
    memset(kinds, 0, n_pixels * 8);
    for (size_t i = 0; i < n_pixels; ++i) {
        const uint32_t x = pixels[i] & 0xffffu, y = pixels[i] >> 16;
        rpt_rng_state r = rng_full[(size_t)y * config->width + x];
        r.n += sample;
        g_kind_log = kinds + 8 * i;
        trace_pixel(x, y, *config, r, sc, cnt);
    }
    g_kind_log = nullptr;
    return 0;
}

/* Analysis hook (tools/node_visit_share.py): how often each BVH node is popped by the nearest-hit and by the any-hit walks of
 * n_samples samples of every pixel — which part of a tree a top-of-tree cache would have to hold. */
int oracle_node_histogram(const rpt_tracing_config *config, const oracle_scene *scene, const rpt_rng_state *rng, uint32_t n_samples,
                          uint64_t *hist_nearest, uint64_t *hist_any) {
    Scene sc = make_scene(scene);
    Counters cnt;
    cnt.hist_nearest = hist_nearest;
    cnt.hist_any = hist_any;
    for (uint32_t y = 0; y < config->height; ++y)
        for (uint32_t x = 0; x < config->width; ++x) {
            rpt_rng_state r = rng[(size_t)y * config->width + x];
            for (uint32_t s = 0; s < n_samples; ++s) {
                trace_pixel(x, y, *config, r, sc, cnt);
                r.n += 1;
            }
        }
    return (int)cnt.error_flags;
}

/* rays[(y*W + x)*6..] = extension ray of bounce `bounce` of sample rng[i] of every pixel; valid[i] = 0 if the path ended earlier */
int oracle_dump_rays(const rpt_tracing_config *config, const oracle_scene *scene, const rpt_rng_state *rng, uint32_t bounce,
                     float *rays, uint8_t *valid) {
    Scene sc = make_scene(scene);
    Counters cnt;
    for (uint32_t y = 0; y < config->height; ++y)
        for (uint32_t x = 0; x < config->width; ++x) {
            size_t i = (size_t)y * config->width + x;
            g_ray_dump = rays + 6 * i;
            g_ray_dump_bounce = bounce;
            g_ray_dump_hit = false;
            trace_pixel(x, y, *config, rng[i], sc, cnt);
            valid[i] = g_ray_dump_hit ? 1 : 0;
        }
    g_ray_dump = nullptr;
    return 0;
}

/* Post-accumulation step: mean = sum / sample_count (src/trace.rs:303-308) followed by display tonemap operator `op`
 * (src/resources/render.wgsl:36-153; 0 none, 1 Reinhard, 2 ACES Narkowicz x0.6, 3 ACES Narkowicz, 4 ACES Hill,
 * 5 Neutral, 6 Uncharted).  WGSL leaves the association of mat*vec and of a*b+c to the implementation; this
 * restatement fixes them as written (left to right, matrix product accumulated column by column). */
static V3 wg_div(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
static V3 wg_sat(V3 a) { return V3{m_min(m_max(a.x, 0.0f), 1.0f), m_min(m_max(a.y, 0.0f), 1.0f), m_min(m_max(a.z, 0.0f), 1.0f)}; }
static V3 wg_aces_narkowicz(V3 x) {
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return wg_sat(wg_div(x * (a * x + splat3(b)), x * (c * x + splat3(d)) + splat3(e)));
}
static V3 wg_aces_hill(V3 x) {
    V3 color = v3(0.59719f, 0.07600f, 0.02840f) * x.x + v3(0.35458f, 0.90834f, 0.13383f) * x.y + v3(0.04823f, 0.01566f, 0.83777f) * x.z;
    V3 a = color * (color + splat3(0.0245786f)) - splat3(0.000090537f);
    V3 b = color * (0.983729f * color + splat3(0.4329510f)) + splat3(0.238081f);
    color = wg_div(a, b);
    color = v3(1.60475f, -0.10208f, -0.00327f) * color.x + v3(-0.53108f, 1.10813f, -0.07276f) * color.y +
            v3(-0.07367f, -0.00605f, 1.07602f) * color.z;
    return wg_sat(color);
}
static V3 wg_curve(V3 x, float a, float b, float c, float d, float e, float f) {
    return wg_div(x * (a * x + splat3(c * b)) + splat3(d * e), x * (a * x + splat3(b)) + splat3(d * f)) - splat3(e / f);
}
int oracle_resolve(const float *accum_rgba, size_t n_pixels, float sample_count, uint32_t op, float *out_rgb) {
    for (size_t i = 0; i < n_pixels; ++i) {
        V3 x = v3(accum_rgba[4 * i] / sample_count, accum_rgba[4 * i + 1] / sample_count, accum_rgba[4 * i + 2] / sample_count);
        V3 r = x;
        switch (op) {
            case 1: r = wg_div(x, x + splat3(1.0f)); break;
            case 2: r = wg_aces_narkowicz(x * 0.6f); break;
            case 3: r = wg_aces_narkowicz(x); break;
            case 4: r = wg_aces_hill(x); break;
            case 5: {
                V3 white_scale = wg_div(splat3(1.0f), wg_curve(splat3(5.3f), 0.2f, 0.29f, 0.24f, 0.272f, 0.02f, 0.3f));
                r = wg_curve(x * white_scale, 0.2f, 0.29f, 0.24f, 0.272f, 0.02f, 0.3f) * white_scale;
                r = wg_div(r, splat3(1.0f));
                break;
            }
            case 6: {
                V3 curr = wg_curve(x * 2.0f, 0.15f, 0.50f, 0.10f, 0.20f, 0.02f, 0.30f);
                V3 white_scale = wg_div(splat3(1.0f), wg_curve(splat3(11.2f), 0.15f, 0.50f, 0.10f, 0.20f, 0.02f, 0.30f));
                r = curr * white_scale;
                break;
            }
            default: break;
        }
        out_rgb[3 * i] = r.x; out_rgb[3 * i + 1] = r.y; out_rgb[3 * i + 2] = r.z;
    }
    return 0;
}

float oracle_lds(uint32_t n, uint32_t dimension, uint32_t offset, uint32_t *out_product) {
    if (out_product) *out_product = LDS_PRIMES[dimension & 31u] * (n + offset);
    return lds(n, dimension, offset);
}

/* op: 0 sin, 1 cos, 2 acos, 3 exp, 4 pow(x,y), 5 asin, 6 atan2(x,y), 7 sqrt, 8 x/y */
/* The reference's two BSDFs that trace_pixel never instantiates (kernels/src/bsdf.rs:46-105 Lambertian, 107-176 Glass;
 * SURVEY.md 8f N4), restated for completeness of the kernels crate: one item = view(3) normal(3) r(3) albedo(3) ior
 * roughness pad(2) -> pdf, lobe (u32 bits), spectrum(3), direction(3).
 * kind 0: Lambertian::sample   1: Glass::sample   2: Lambertian::{evaluate, pdf} with sample_direction = r
 * kind 3: Glass::{evaluate, pdf} with lobe = (u32) r.x */
int oracle_bsdf(int kind, size_t n, const float *in, float *out) {
    auto cartesian = [](V3 up, V3 &right, V3 &forward) {                     /* util.rs:34-40 */
        V3 temp_vec = normalize(cross(up, v3(0.1f, 0.5f, 0.9f)));
        right = normalize(cross(temp_vec, up));
        forward = normalize(cross(up, right));
    };
    for (size_t i = 0; i < n; ++i) {
        const float *p = in + 16 * i;
        float *o = out + 8 * i;
        V3 view = xyz(p), normal = xyz(p + 3), r = xyz(p + 6), albedo = xyz(p + 9);
        float ior = p[12], roughness = p[13];
        float pdf = 0.0f;
        uint32_t lobe = 0u;
        V3 spectrum = splat3(0.0f), dir = splat3(0.0f);
        if (kind == 0) {                                                     /* bsdf.rs:71-92 */
            V3 nt, nb;
            cartesian(normal, nt, nb);
            float theta = m_acos(m_sqrt(r.x)), phi = 2.0f * PI_F * r.y;      /* util.rs:24-32 */
            V3 s = v3(m_sin(theta) * m_cos(phi), m_cos(theta), m_sin(theta) * m_sin(phi));
            dir = normalize(v3(s.x * nb.x + s.y * normal.x + s.z * nt.x, s.x * nb.y + s.y * normal.y + s.z * nt.y,
                               s.x * nb.z + s.y * normal.z + s.z * nt.z));
            float cos_theta = m_max(dot(normal, dir), 0.0f);
            pdf = cos_theta / PI_F;
            spectrum = albedo / PI_F * cos_theta;
            lobe = 0u;
        } else if (kind == 1) {                                              /* bsdf.rs:130-168 */
            bool inside = dot(normal, view) < 0.0f;
            V3 nrm = inside ? -normal : normal;
            float in_ior = inside ? ior : 1.0f, out_ior = inside ? 1.0f : ior;
            float a_g = roughness * roughness;                               /* util.rs:117-142 */
            float theta_m = m_atan((a_g * m_sqrt(r.x)) / m_sqrt(1.0f - r.x));
            float phi_m = 2.0f * PI_F * r.y;
            V3 m = v3(m_sin(theta_m) * m_cos(phi_m), m_cos(theta_m), m_sin(theta_m) * m_sin(phi_m));
            V3 nt, nb;
            cartesian(nrm, nt, nb);
            V3 mn = normalize(v3(m.x * nb.x + m.y * nrm.x + m.z * nt.x, m.x * nb.y + m.y * nrm.y + m.z * nt.y,
                                 m.x * nb.z + m.y * nrm.z + m.z * nt.z));
            float f0 = (in_ior - out_ior) / (in_ior + out_ior);             /* util.rs:233-236 */
            f0 = f0 * f0;
            float fresnel = f0 + (1.0f - f0) * rptm::powi5(1.0f - m_max(dot(mn, view), 0.0f));
            pdf = 1.0f;
            if (r.z <= fresnel) {
                dir = normalize(2.0f * rptm::absr(dot(view, mn)) * mn - view);
                lobe = 1u;
                spectrum = splat3(1.0f);
            } else {
                float eta = in_ior / out_ior;
                float c = dot(view, mn);
                float d = dot(view, nrm);
                float sg = d != d ? d : ((rptm::f2u(d) >> 31) ? -1.0f : 1.0f);        /* f32::signum */
                dir = normalize((eta * c - sg * m_sqrt(m_max(1.0f + eta * (c * c - 1.0f), 0.0f))) * mn - eta * view);
                lobe = 3u;
                spectrum = albedo;
            }
        } else if (kind == 2) {                                              /* bsdf.rs:59-69, 94-104 */
            float cos_theta = m_max(dot(normal, r), 0.0f);
            pdf = cos_theta / PI_F;
            spectrum = albedo / PI_F * cos_theta;
        } else {                                                             /* bsdf.rs:115-128, 170-176 */
            lobe = rptm::f2u32_sat(r.x);
            spectrum = lobe == 1u ? splat3(1.0f) : albedo;
            pdf = 1.0f;
        }
        o[0] = pdf; o[1] = rptm::u2f(lobe);
        o[2] = spectrum.x; o[3] = spectrum.y; o[4] = spectrum.z;
        o[5] = dir.x; o[6] = dir.y; o[7] = dir.z;
    }
    return 0;
}

int oracle_math(int op, const float *x, const float *y, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        float r;
        switch (op) {
            case 0: r = m_sin(x[i]); break;
            case 1: r = m_cos(x[i]); break;
            case 2: r = m_acos(x[i]); break;
            case 3: r = m_exp(x[i]); break;
            case 10: r = m_exp_sky(x[i]); break;
            case 4: r = m_pow(x[i], y[i]); break;
            case 5: r = m_asin(x[i]); break;
            case 6: r = m_atan2(x[i], y[i]); break;
            case 7: r = m_sqrt(x[i]); break;
            case 8: r = x[i] / y[i]; break;
            default: return -1;
        }
        out[i] = r;
    }
    return 0;
}

/* Sky model on a batch of directions (origin fixed), for unit tests. */
int oracle_sky(const float *sun_direction4, const float *origin3, const float *dirs, float *out_rgb, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        V3 c = sky_scatter(sun_direction4, xyz(origin3), xyz(dirs + 3 * i));
        out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    }
    return 0;
}

}  // extern "C"
