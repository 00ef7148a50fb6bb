/*
 * shadow_order.h — the visiting order of the any-hit (shadow) walks and of the hit-or-miss lanes of the last extension rays, decided once per scene at
 * upload by PROBE RAYS.  One core (host + device functions over the uploaded reference buffers), two drivers: kernels (what rpt_upload_scene runs since
 * round 6) and a sequential host loop (rpt_debug_*_order_host: the checker — same rays, same node visits, same decision, asserted in tests).
 *
 * What the reference fixes and what it leaves free.  A shadow query (kernels/src/light_pick.rs:141-148) is intersect_any
 * (intersection.rs:173-175): intersect_front_to_back<false>, of whose result ONLY `.hit` is read (light_pick.rs:148).  Inside that walk
 * `result.t` stays 1 000 000.0 until the first triangle is accepted, every box is tested against that constant (:212-213), and the first
 * accepted triangle returns (:201-203).  So whether a node is entered depends on the boxes of its ancestors alone — not on the order
 * siblings are visited in — and `.hit` is the OR of the accept test (:191-194 with t <= max_t) over the triangles of the reachable leaves.
 * ANY visiting order gives the reference's `.hit` bit for bit (tests/test_anyhit_order.py: left-first, right-first, far-first, random
 * per (ray, node) and breadth-first against intersect_front_to_back<false> on 10^6 rays).  Nearest-hit walks have no such freedom: their
 * order decides ties in t (k_traverse.h header).
 *
 * What the freedom is worth (tools/anyhit_order_sim.py, profiles/r05_anyhit_order_sim.txt): an UNOCCLUDED ray visits the same nodes under
 * every order; an occluded ray stops at the first occluder it meets, and how soon that is depends on the scene.  Replayed on the real
 * shadow rays of the streamed kernels: DarkCornell — near-first finds the occluder after 28.9 node visits of the 30.6 an unoccluded ray
 * makes (a ray leaves a wall: the boxes around its origin are nearest and hold its own surface), a FIXED order that enters the more
 * OPAQUE child first (triangle surface / box surface of the subtree) after 14.1: - 28 % wave-instructions per shadow ray; VeachMIS (most
 * occluders are the far side of the very light sphere a ray aims at) near-first 16.2, opaque-first 20.9: + 7 %.  Packet descent
 * (one stack of (node, lane mask) per wave) needs + 98 % / + 1 245 %, a stackless threaded layout + 9 % / + 12 %, a wave vote - 2 % / + 7 %.
 * (With the rays that decide nothing elided — k_shade.h, half of all NEE evaluations — the same replay gives - 15 % on DarkCornell, - 3 % on VeachMIS.)
 * Neither order wins everywhere, so the library measures: at upload it throws up to SHADOW_PROBE_RAYS synthetic shadow rays (surface points
 * chosen by area on the non-emissive triangles, light points through the scene's own light-pick table as light_pick.rs:8-23 draws them, kept when
 * the light faces the point and is above its horizon — the rays the device walks) through both orders and counts node visits.  If opaque-first
 * needs fewer than SHADOW_FIXED_GAIN of near-first's, the shadow kernels walk a copy of the tree whose child pairs are flipped so that the preferred
 * child sits in the left slot, in fixed left-first order (no `tl > tr`, no swap); otherwise they keep the reference's near-first order.
 * Deterministic (ray i draws from its own generator), and whatever it decides the image is the same.  RPT_SHADOW_ORDER=near|fixed overrides (tests
 * run every NEE case both ways; the library's knobs are read in one place, rpt_ctx.h rpt_knobs).
 *
 * Round 6: the probes run as kernels.  On a 1 M-triangle scene the sequential probe was 17 - 40 ms of rpt_upload_scene — triangle areas, the
 * per-node sums, a cumulative distribution over all triangles, 8 192 walks through cold memory.  Now: per-node sums level by level on the device
 * (leaves first, an inner node once both children are done: tree depth + 1 small launches), surface points drawn by DESCENDING the tree with those sums
 * (no O(triangles) distribution), one thread per probe ray with its stack in LDS, four counters back.  Every quantity that reaches the decision is a chain
 * of IEEE float / double operations in a fixed order (no contraction; correctly rounded division and square root on both sides), so the kernels and the
 * host loop agree to the last node visit.
 */
#ifndef RPT_SHADOW_ORDER_H
#define RPT_SHADOW_ORDER_H

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rpt/shared_structs.h"
#include "rpt_math.h"

#define SHADOW_PROBE_RAYS 4096     /* candidates; about half survive the "decides something" filter */
#define SHADOW_FIXED_GAIN 0.95
#define LAST_PROBE_RAYS 1024
#define ORDER_PROBE_STACK 32       /* a validated pool is at most 31 levels deep (rpt_hip.hip validate_scene; intersection.rs:178) */

struct ShadowOrder {
    bool fixed = false;                 /* walk the flipped tree left-first */
    std::vector<uint8_t> flip;          /* per child pair p = nodes (2p + 1, 2p + 2): the right child is the preferred one */
    double visits_near = 0.0, visits_fixed = 0.0;   /* node visits per probe ray under either order */
    uint32_t probe_rays = 0, probe_occluded = 0;
    double probe_ms = 0.0;              /* time of the whole decision (host clock around it) */
    const char *why = "no lights";
};

/* ---- the hit-or-miss lanes of the last extension rays (k_traverse.h k_traverse_nearest_stream LAST) -------------------------------------------------
 * Without NEE the last extension ray of a path that cannot end on an emitter only has to say "hit or miss": the part of the reference's walk up to its first
 * accepted triangle, which is an any-hit walk (result.t is 1e6 throughout) and as free in its order as a shadow query.  These rays are not shadow rays — they
 * leave a surface in a direction the BSDF drew, and in a closed scene all of them hit — so the order is chosen on rays of their kind: points by area on the
 * non-emissive triangles, cosine-distributed directions about the shading normal, walked near child first (the primary image: no second copy
 * needed) and in fixed order under three rules that put into the left slot the child that is (1) more opaque, (2) the smaller subtree, (3) more opaque per
 * node of its subtree — the classic "most likely per unit of cost first" for a search that stops at its first success.  tools/last_bounce_sim.py and the
 * replay of the real bounce-3 rays of DarkCornell: 18.9 node visits near first, 15.2 / 12.2 / 11.9 under the three rules (the whole walk: 25.8).  LDS-image scenes only
 * (the only ones with a LAST kernel). */
struct LastOrder {
    int rule = 0;                       /* 0: near child first on the primary image; 1..3: fixed order over a copy flipped by that rule */
    std::vector<uint8_t> flip;
    double visits[4] = {0.0, 0.0, 0.0, 0.0};   /* node visits per probe ray: near first, rules 1..3 */
    uint32_t probe_rays = 0, probe_hits = 0;
    double probe_ms = 0.0;
};

namespace order_probe {

struct float4_like { float x, y, z, w; };       /* (float4 without <hip/hip_runtime.h>: this header is plain C++ for the host driver) */
struct V { float x, y, z; };
RPT_HD V sub(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
RPT_HD V cross(V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RPT_HD float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RPT_HD V vtx(const rpt_per_vertex_data &p) { return V{p.vertex[0], p.vertex[1], p.vertex[2]}; }
RPT_HD float absf(float x) { return rptm::absr(x); }

/* the scene as uploaded + the per-node sums of the probe (host vectors or device arrays) */
struct View {
    const rpt_per_vertex_data *pv;
    const rpt_triangle *idx;
    const rpt_bvh_node *nodes;
    const rpt_material_data *mats;
    const rpt_light_pick_entry *lp;
    uint32_t nt, nn, nlp;
    const double *tri_area;       /* 0.5 |e1 x e2| per triangle */
    const double *area_all;       /* triangle surface below every node */
    const double *area_ne;        /* ... of the triangles whose material does not emit */
    const double *count;          /* nodes of every subtree */
    uint32_t sub, lanes;          /* device driver: this thread is lane `sub` of the `lanes` (a power of two, adjacent lanes of one wave) that walk one ray together — they
                                     share every decision and split the triangles of a leaf; host driver: 0, 1 */
    const float4_like *geom;      /* device driver: 3 x float4 per triangle — a, e1 = b - a, e2 = c - a as k_derive_triangles computed them (the same subtractions
                                     tri() makes: one 48-byte record instead of the index record and three 64-byte vertices); host driver: null */
};

RPT_HD bool emissive(const View &s, uint32_t t) {
    const float *e = s.mats[s.idx[t].material].emissive;
    return e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f;
}
/* |(b - a) x (c - a)|^2 of a triangle, as rpt_hip.hip k_derive_triangles hands it out */
RPT_HD float triangle_cross_sq(const rpt_per_vertex_data *pv, const rpt_triangle &t) {
    const V a = vtx(pv[t.v0]), e1 = sub(vtx(pv[t.v1]), a), e2 = sub(vtx(pv[t.v2]), a);
    const float cx = e1.y * e2.z - e1.z * e2.y, cy = e1.z * e2.x - e1.x * e2.z, cz = e1.x * e2.y - e1.y * e2.x;
    return (cx * cx + cy * cy) + cz * cz;
}
RPT_HD double area_of_cross_sq(float cross_sq) { return 0.5 * (double)rptm::sqrtr(cross_sq); }
RPT_HD double opacity(const View &s, uint32_t n) {
    const rpt_bvh_node &b = s.nodes[n];
    const double ex = (double)b.aabb_max[0] - (double)b.aabb_min[0], ey = (double)b.aabb_max[1] - (double)b.aabb_min[1], ez = (double)b.aabb_max[2] - (double)b.aabb_min[2];
    double half = ex * ey + ey * ez + ez * ex;
    if (!(half > 1e-30)) half = 1e-30;
    const double o = s.area_all[n] / half;
    return o < 1.0 ? o : 1.0;
}
/* the right child of pair p is the preferred one under rule 1 (more opaque), 2 (smaller subtree), 3 (more opaque per node of its subtree) */
RPT_HD bool prefers_right(const View &s, uint32_t p, int rule) {
    const uint32_t L = 2u * p + 1u, R = L + 1u;
    if (rule == 2) return s.count[R] < s.count[L];
    const double ol = opacity(s, L), orr = opacity(s, R);
    return rule == 1 ? orr > ol : orr / s.count[R] > ol / s.count[L];
}

/* the box test of the walk with prev_min_t = 1e6 (plain float code with a reciprocal direction: this is a cost estimate, not a result) */
RPT_HD bool box(const rpt_bvh_node &n, V o, V id, float &tmin) {
    float tx1 = (n.aabb_min[0] - o.x) * id.x, tx2 = (n.aabb_max[0] - o.x) * id.x;
    float lo = __builtin_fminf(tx1, tx2), hi = __builtin_fmaxf(tx1, tx2);
    float ty1 = (n.aabb_min[1] - o.y) * id.y, ty2 = (n.aabb_max[1] - o.y) * id.y;
    lo = __builtin_fmaxf(lo, __builtin_fminf(ty1, ty2)); hi = __builtin_fminf(hi, __builtin_fmaxf(ty1, ty2));
    float tz1 = (n.aabb_min[2] - o.z) * id.z, tz2 = (n.aabb_max[2] - o.z) * id.z;
    lo = __builtin_fmaxf(lo, __builtin_fminf(tz1, tz2)); hi = __builtin_fminf(hi, __builtin_fmaxf(tz1, tz2));
    tmin = lo;
    return hi >= lo && hi > 0.0f && lo < 1000000.0f;
}
RPT_HD bool tri_test(V a, V e1, V e2, V o, V d, float max_t) {
    V p = cross(d, e2);
    float det = dot(e1, p);
    if (absf(det) < 1e-6f) return false;
    float inv = 1.0f / det;
    V tv = sub(o, a);
    float u = dot(tv, p) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    V q = cross(tv, e1);
    float v = dot(d, q) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    float tt = dot(e2, q) * inv;
    return tt > 0.001f && tt <= max_t;
}
/* does the ray pass any triangle of a leaf?  (device driver: the records of eight triangles are in flight before the first is tested — one thread walks
 * a ray, and a leaf of the clustered stand-in holds sixty triangles: tested one load at a time the probe was 8 ms of memory latency) */
RPT_HD bool leaf_test(const View &s, uint32_t first, uint32_t count, V o, V d, float max_t) {
    if (s.geom) {
        bool found = false;
        for (uint32_t k0 = s.sub; k0 < count && !found; k0 += 4u * s.lanes) {
            float4_like g[12];
            RPT_UNROLL
            for (uint32_t j = 0; j < 4u; ++j)
                if (k0 + j * s.lanes < count) { const float4_like *r = s.geom + 3u * (size_t)(first + k0 + j * s.lanes); g[3 * j] = r[0]; g[3 * j + 1] = r[1]; g[3 * j + 2] = r[2]; }
            RPT_UNROLL
            for (uint32_t j = 0; j < 4u; ++j)
                if (k0 + j * s.lanes < count && tri_test(V{g[3 * j].x, g[3 * j].y, g[3 * j].z}, V{g[3 * j + 1].x, g[3 * j + 1].y, g[3 * j + 1].z}, V{g[3 * j + 2].x, g[3 * j + 2].y, g[3 * j + 2].z}, o, d, max_t))
                    found = true;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (s.lanes > 1u) {          /* the lanes of a ray walk in lockstep: whoever found a triangle tells the others */
            const uint32_t lane = __lane_id();
            found = ((__builtin_amdgcn_ballot_w64(found) >> (lane & ~(s.lanes - 1u))) & ((1ull << s.lanes) - 1ull)) != 0ull;
        }
#endif
        return found;
    }
    for (uint32_t k = 0; k < count; ++k) {
        const rpt_triangle &t = s.idx[first + k];
        const V a = vtx(s.pv[t.v0]);
        if (tri_test(a, sub(vtx(s.pv[t.v1]), a), sub(vtx(s.pv[t.v2]), a), o, d, max_t)) return true;
    }
    return false;
}

/* ray i of a probe draws from its own generator (the kernels run one thread per ray) */
struct Rng {
    uint64_t s;
    RPT_HD float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffffu) * (1.0f / 16777216.0f); }
};
RPT_HD Rng ray_rng(uint64_t seed, uint32_t i) {
    Rng r{seed + (uint64_t)(i + 1u) * 0x9e3779b97f4a7c15ull};
    r.next(); r.next();
    return r;
}

/* a triangle with probability proportional to its weight — its area, 0 for emissive ones unless `all` — by descending the tree with the per-node sums:
 * at an inner node left with probability sum(L) / sum(n), in a leaf along its triangles.  O(depth), no distribution over all triangles. */
RPT_HD uint32_t pick_triangle(const View &s, bool all, float u) {
    const double *sum = all ? s.area_all : s.area_ne;
    uint32_t n = 0u;
    double x = (double)u * sum[0];
    while (s.nodes[n].triangle_count == 0u) {
        const uint32_t L = s.nodes[n].left_or_first;
        if (x < sum[L]) n = L;
        else { x -= sum[L]; n = L + 1u; }
    }
    const uint32_t first = s.nodes[n].left_or_first, cnt = s.nodes[n].triangle_count;
    uint32_t pick = first, last_weighted = 0xffffffffu;
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t t = first + k;
        const double w = (!all && emissive(s, t)) ? 0.0 : s.tri_area[t];
        if (w > 0.0) {
            last_weighted = t;
            if (x < w) return t;
            x -= w;
        }
    }
    return last_weighted != 0xffffffffu ? last_weighted : pick;        /* (rounding walked past the end of the leaf) */
}
RPT_HD V point_on(const View &s, uint32_t t, float r1, float r2) {
    V a = vtx(s.pv[s.idx[t].v0]), b = vtx(s.pv[s.idx[t].v1]), c = vtx(s.pv[s.idx[t].v2]);
    const float sq = rptm::sqrtr(r1), wa = 1.0f - sq, wb = sq * (1.0f - r2), wc = sq * r2;
    return V{wa * a.x + wb * b.x + wc * c.x, wa * a.y + wb * b.y + wc * c.y, wa * a.z + wb * b.z + wc * c.z};
}
RPT_HD V normal_sum(const View &s, uint32_t t) {
    const float *a = s.pv[s.idx[t].v0].normal, *b = s.pv[s.idx[t].v1].normal, *c = s.pv[s.idx[t].v2].normal;
    return V{a[0] + b[0] + c[0], a[1] + b[1] + c[1], a[2] + b[2] + c[2]};
}

/* probe ray i of the shadow probe: surface point by area (non-emissive triangles; all of them when everything emits), light point as pick_light +
 * pick_triangle_point draw it (light_pick.rs:8-23, 100-134); false: no ray (a degenerate pair of points, or a ray the device does not walk) */
RPT_HD bool shadow_probe_ray(const View &s, bool all, uint32_t i, V &o, V &d, float &max_t) {
    Rng rng = ray_rng(0x9e3779b97f4a7c15ull, i);
    const uint32_t t0 = pick_triangle(s, all, rng.next());
    const float p1 = rng.next(), p2 = rng.next();
    const V p = point_on(s, t0, p1, p2);
    uint32_t e_at = (uint32_t)(rng.next() * (float)s.nlp);
    if (e_at >= s.nlp) e_at = s.nlp - 1u;
    const rpt_light_pick_entry &e = s.lp[e_at];
    const uint32_t lt = rng.next() < e.ratio ? e.triangle_index_a : e.triangle_index_b;
    const float q1 = rng.next(), q2 = rng.next();
    if (lt >= s.nt) return false;
    const V q = point_on(s, lt, q1, q2);
    d = sub(q, p);
    const float dist = rptm::sqrtr(dot(d, d));
    if (!(dist > 1e-4f)) return false;
    d = V{d.x / dist, d.y / dist, d.z / dist};
    /* only rays the device walks: a light point that faces away (light_pdf = 0) or lies below the surface's horizon (bsdf_pdf = 0) adds a zero
     * term whatever its shadow ray finds, and k_shade does not queue that ray (about half of all NEE evaluations) */
    if (!(dot(normal_sum(s, t0), d) > 0.0f) || !(dot(normal_sum(s, lt), d) < 0.0f)) return false;
    o = V{p.x + d.x * 0.001f, p.y + d.y * 0.001f, p.z + d.z * 0.001f};          /* light_pick.rs:141-147, EPS = 0.001 */
    max_t = dist - 0.002f;
    return true;
}
/* probe ray i of the last-bounce probe: a point by area on the non-emissive triangles, a cosine-distributed direction about its shading normal */
RPT_HD bool last_probe_ray(const View &s, uint32_t i, V &o, V &d) {
    Rng rng = ray_rng(0x2545f4914f6cdd1dull, i);
    const uint32_t t0 = pick_triangle(s, false, rng.next());
    const float p1 = rng.next(), p2 = rng.next();
    const V p = point_on(s, t0, p1, p2);
    V n = normal_sum(s, t0);
    const float nl = rptm::sqrtr(dot(n, n));
    const float u1 = rng.next(), u2 = rng.next();
    if (!(nl > 1e-12f)) return false;
    n = V{n.x / nl, n.y / nl, n.z / nl};
    const V h = absf(n.x) < 0.5f ? V{1.0f, 0.0f, 0.0f} : V{0.0f, 1.0f, 0.0f};
    V t = cross(n, h);
    const float tl = rptm::sqrtr(dot(t, t));
    t = V{t.x / tl, t.y / tl, t.z / tl};
    const V bt = cross(n, t);
    float sn, cs;
    rptm::sincosr(6.2831853f * u2, sn, cs);
    const float r = rptm::sqrtr(u1), x = r * cs, y = r * sn, z = rptm::sqrtr(__builtin_fmaxf(0.0f, 1.0f - u1));
    d = V{x * t.x + y * bt.x + z * n.x, x * t.y + y * bt.y + z * n.y, x * t.z + y * bt.z + z * n.z};
    if (d.x == 0.0f || d.y == 0.0f || d.z == 0.0f) return false;
    o = V{p.x + n.x * 0.001f, p.y + n.y * 0.001f, p.z + n.z * 0.001f};
    return true;
}

/* node visits of one any-hit walk; FIXED: the preferred child of every pair first (flip: per pair, nullable = left), else near first.
 * `stack(k)` is entry k of the walk's own stack (host: an array; device: LDS, one column per lane). */
template <bool FIXED, class Stack>
RPT_HD uint32_t walk(const View &s, const uint8_t *flip, V o, V d, float max_t, Stack stack, bool &occluded) {
    int sp = 0;
    uint32_t visits = 0, node = 0;
    occluded = false;
    const V id{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    for (;;) {
        visits += 1;
        const rpt_bvh_node &n = s.nodes[node];
        bool descend = false;
        if (n.triangle_count != 0u) {
            if (leaf_test(s, n.left_or_first, n.triangle_count, o, d, max_t)) { occluded = true; return visits; }
        } else {
            const uint32_t L = n.left_or_first, R = L + 1u;
            float tl, tr;
            const bool hl = box(s.nodes[L], o, id, tl), hr = box(s.nodes[R], o, id, tr);
            const bool right = FIXED ? (hr && (!hl || flip[L >> 1] != 0)) : (hr && (!hl || tl > tr));
            if (hl || hr) {
                if (hl && hr && sp < ORDER_PROBE_STACK) stack(sp++) = right ? L : R;
                node = right ? R : L;
                descend = true;
            }
        }
        if (!descend) {
            if (sp == 0) return visits;
            node = stack(--sp);
        }
    }
}

struct HostStack {
    uint32_t *e;
    uint32_t &operator()(int k) const { return e[k]; }
};

/* per-node sums on the host: children before parents (explicit post-order: the pool's index order is the builder's business) */
inline void host_sums(const rpt_per_vertex_data *pv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn, const rpt_material_data *mats,
                      const float *cross_sq, std::vector<double> &tri_area, std::vector<double> &area_all, std::vector<double> &area_ne, std::vector<double> &count) {
    tri_area.assign(nt, 0.0); area_all.assign(nn, 0.0); area_ne.assign(nn, 0.0); count.assign(nn, 1.0);
    for (size_t t = 0; t < nt; ++t) tri_area[t] = area_of_cross_sq(cross_sq ? cross_sq[t] : triangle_cross_sq(pv, idx[t]));
    std::vector<uint32_t> order;
    order.reserve(nn);
    std::vector<uint32_t> st(1, 0u);
    while (!st.empty()) {
        uint32_t n = st.back(); st.pop_back();
        order.push_back(n);
        if (nodes[n].triangle_count == 0u) { st.push_back(nodes[n].left_or_first); st.push_back(nodes[n].left_or_first + 1u); }
    }
    for (size_t i = order.size(); i-- > 0;) {
        const uint32_t at = order[i];
        const rpt_bvh_node &n = nodes[at];
        if (n.triangle_count == 0u) {
            area_all[at] = area_all[n.left_or_first] + area_all[n.left_or_first + 1u];
            area_ne[at] = area_ne[n.left_or_first] + area_ne[n.left_or_first + 1u];
            count[at] = 1.0 + count[n.left_or_first] + count[n.left_or_first + 1u];
        } else {
            for (uint32_t k = 0; k < n.triangle_count; ++k) {
                const uint32_t t = n.left_or_first + k;
                const float *e = mats[idx[t].material].emissive;
                area_all[at] += tri_area[t];
                if (!(e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f)) area_ne[at] += tri_area[t];
            }
        }
    }
}

/* the decisions, from the counters of either driver */
inline void decide_shadow(ShadowOrder &so, uint64_t visits_near, uint64_t visits_fixed, uint32_t rays, uint32_t occluded, int forced /* RPT_SHADOW_ORDER: 0 near, 1 fixed, -1 the probe's */) {
    so.probe_rays = rays;
    so.probe_occluded = occluded;
    if (rays == 0) { so.why = "no probe ray could be formed"; return; }
    so.visits_near = (double)visits_near / rays;
    so.visits_fixed = (double)visits_fixed / rays;
    so.fixed = so.visits_fixed < SHADOW_FIXED_GAIN * so.visits_near;
    so.why = so.fixed ? "opaque-first needs fewer node visits on the probe rays" : "near-first needs no more node visits on the probe rays";
    if (forced == 1) { so.fixed = true; so.why = "RPT_SHADOW_ORDER=fixed"; }
    else if (forced == 0) { so.fixed = false; so.why = "RPT_SHADOW_ORDER=near"; }
}
inline void decide_last(LastOrder &lo, const uint64_t v[4], uint32_t rays, uint32_t hits, int forced /* RPT_LAST_ORDER: 0 near, 1 opaque, 2 small, 3 ratio; -1 (and 4 = off, decided elsewhere): the probe's */) {
    lo.probe_rays = rays;
    lo.probe_hits = hits;
    if (rays == 0) return;
    for (int q = 0; q < 4; ++q) lo.visits[q] = (double)v[q] / rays;
    int best = 1;
    for (int q = 2; q <= 3; ++q) if (lo.visits[q] < lo.visits[best]) best = q;
    lo.rule = lo.visits[best] < SHADOW_FIXED_GAIN * lo.visits[0] ? best : 0;
    if (forced >= 0 && forced <= 3) lo.rule = forced;              /* tests and A/B */
}

struct Clock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace order_probe

/* Host driver.  Expects a validated scene (rpt_hip.hip validate_scene: links in range, no cycles, leaf ranges inside the index buffer).  `pair_shaped`: children of
 * every inner node are the nodes (2p + 1, 2p + 2) of one pair — what the flipped copies can express; otherwise near-first stays. */
inline ShadowOrder choose_shadow_order(const rpt_per_vertex_data *pv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn,
                                       const rpt_material_data *mats, const rpt_light_pick_entry *lp, size_t nlp, bool pair_shaped,
                                       const float *cross_sq = nullptr /* optional: |(b - a) x (c - a)|^2 per triangle */, int forced = -1) {
    using namespace order_probe;
    ShadowOrder so;
    const Clock clock;
    if (nlp == 0 || lp[0].ratio < 0.0f || nt == 0) { so.probe_ms = clock.ms(); return so; }
    if (!pair_shaped || nn < 3) { so.why = "node pool is not pair-shaped"; so.probe_ms = clock.ms(); return so; }
    std::vector<double> tri_area, area_all, area_ne, count;
    host_sums(pv, idx, nt, nodes, nn, mats, cross_sq, tri_area, area_all, area_ne, count);
    const View s{pv, idx, nodes, mats, lp, (uint32_t)nt, (uint32_t)nn, (uint32_t)nlp, tri_area.data(), area_all.data(), area_ne.data(), count.data(), 0u, 1u, nullptr};
    so.flip.assign((nn - 1) / 2, 0);
    for (size_t p = 0; p < so.flip.size(); ++p) so.flip[p] = prefers_right(s, (uint32_t)p, 1) ? 1 : 0;
    const bool all = !(area_ne[0] > 0.0);
    if (all && !(area_all[0] > 0.0)) { so.why = "degenerate geometry"; so.probe_ms = clock.ms(); return so; }
    uint64_t vn = 0, vf = 0;
    uint32_t rays = 0, occluded = 0;
    uint32_t stack[ORDER_PROBE_STACK];
    for (uint32_t i = 0; i < SHADOW_PROBE_RAYS; ++i) {
        V o, d;
        float max_t;
        if (!shadow_probe_ray(s, all, i, o, d, max_t)) continue;
        bool occ_n = false, occ_f = false;
        vn += walk<false>(s, so.flip.data(), o, d, max_t, HostStack{stack}, occ_n);
        vf += walk<true>(s, so.flip.data(), o, d, max_t, HostStack{stack}, occ_f);
        rays += 1;
        occluded += occ_n ? 1u : 0u;
    }
    decide_shadow(so, vn, vf, rays, occluded, forced);
    so.probe_ms = clock.ms();
    return so;
}

inline LastOrder choose_last_order(const rpt_per_vertex_data *pv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn,
                                   const rpt_material_data *mats, bool pair_shaped, int forced = -1) {
    using namespace order_probe;
    LastOrder lo;
    const Clock clock;
    if (!pair_shaped || nn < 3 || nt == 0) { lo.probe_ms = clock.ms(); return lo; }
    std::vector<double> tri_area, area_all, area_ne, count;
    host_sums(pv, idx, nt, nodes, nn, mats, nullptr, tri_area, area_all, area_ne, count);
    const View s{pv, idx, nodes, mats, nullptr, (uint32_t)nt, (uint32_t)nn, 0u, tri_area.data(), area_all.data(), area_ne.data(), count.data(), 0u, 1u, nullptr};
    const size_t P = (nn - 1) / 2;
    std::vector<uint8_t> flips[4];
    for (int r = 1; r <= 3; ++r) {
        flips[r].assign(P, 0);
        for (size_t p = 0; p < P; ++p) flips[r][p] = prefers_right(s, (uint32_t)p, r) ? 1 : 0;
    }
    if (!(area_ne[0] > 0.0)) { lo.probe_ms = clock.ms(); return lo; }
    uint64_t v[4] = {0, 0, 0, 0};
    uint32_t rays = 0, hits = 0;
    uint32_t stack[ORDER_PROBE_STACK];
    for (uint32_t i = 0; i < LAST_PROBE_RAYS; ++i) {
        V o, d;
        if (!last_probe_ray(s, i, o, d)) continue;
        bool hit = false;
        v[0] += walk<false>(s, nullptr, o, d, 1000000.0f, HostStack{stack}, hit);
        for (int q = 1; q <= 3; ++q) v[q] += walk<true>(s, flips[q].data(), o, d, 1000000.0f, HostStack{stack}, hit);
        rays += 1;
        hits += hit ? 1u : 0u;
    }
    decide_last(lo, v, rays, hits, forced);
    if (lo.rule != 0) lo.flip = flips[lo.rule];
    lo.probe_ms = clock.ms();
    return lo;
}

/* the node pool with the children of every flipped pair exchanged (contents move, pair positions stay: links into pairs remain valid) */
inline std::vector<rpt_bvh_node> flipped_nodes(const rpt_bvh_node *nodes, size_t nn, const std::vector<uint8_t> &flip) {
    std::vector<rpt_bvh_node> out(nodes, nodes + nn);
    for (size_t p = 0; p < flip.size(); ++p)
        if (flip[p]) std::swap(out[2 * p + 1], out[2 * p + 2]);
    return out;
}

#endif /* RPT_SHADOW_ORDER_H */
