/*
 * shadow_order.h — the visiting order of the any-hit (shadow) walks, decided once per scene at upload (host code of librpt_hip.so).
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
 * the light faces the point and is above its horizon — the rays the device walks) through both orders on the host and counts node visits.  If opaque-first needs fewer than SHADOW_FIXED_GAIN of near-first's, the shadow
 * kernels walk a copy of the tree whose child pairs are flipped so that the preferred child sits in the left slot, in fixed left-first
 * order (no `tl > tr`, no swap); otherwise they keep the reference's near-first order.  Deterministic (fixed seed), a few milliseconds, and
 * whatever it decides the image is the same.  RPT_SHADOW_ORDER=near|fixed overrides (tests run every NEE case both ways).
 */
#ifndef RPT_SHADOW_ORDER_H
#define RPT_SHADOW_ORDER_H

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rpt/shared_structs.h"

#define SHADOW_PROBE_RAYS 4096     /* candidates; about half survive the "decides something" filter */
#define SHADOW_FIXED_GAIN 0.95

struct ShadowOrder {
    bool fixed = false;                 /* walk the flipped tree left-first */
    std::vector<uint8_t> flip;          /* per child pair p = nodes (2p + 1, 2p + 2): the right child is the preferred one */
    double visits_near = 0.0, visits_fixed = 0.0;   /* node visits per probe ray under either order */
    uint32_t probe_rays = 0, probe_occluded = 0;
    double probe_ms = 0.0;              /* host time of the whole decision */
    const char *why = "no lights";
};

namespace shadow_order_detail {

struct V { float x, y, z; };
inline V sub(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V cross(V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V vtx(const rpt_per_vertex_data &p) { return V{p.vertex[0], p.vertex[1], p.vertex[2]}; }

/* the box test of the walk with prev_min_t = 1e6 (plain float code with a reciprocal direction: this is a cost estimate, not a result) */
inline bool box(const rpt_bvh_node &n, V o, V id, float &tmin) {
    float tx1 = (n.aabb_min[0] - o.x) * id.x, tx2 = (n.aabb_max[0] - o.x) * id.x;
    float lo = std::fmin(tx1, tx2), hi = std::fmax(tx1, tx2);
    float ty1 = (n.aabb_min[1] - o.y) * id.y, ty2 = (n.aabb_max[1] - o.y) * id.y;
    lo = std::fmax(lo, std::fmin(ty1, ty2)); hi = std::fmin(hi, std::fmax(ty1, ty2));
    float tz1 = (n.aabb_min[2] - o.z) * id.z, tz2 = (n.aabb_max[2] - o.z) * id.z;
    lo = std::fmax(lo, std::fmin(tz1, tz2)); hi = std::fmin(hi, std::fmax(tz1, tz2));
    tmin = lo;
    return hi >= lo && hi > 0.0f && lo < 1000000.0f;
}

inline bool tri(const rpt_per_vertex_data *pv, const rpt_triangle &t, V o, V d, float max_t) {
    V a = vtx(pv[t.v0]), e1 = sub(vtx(pv[t.v1]), a), e2 = sub(vtx(pv[t.v2]), a);
    V p = cross(d, e2);
    float det = dot(e1, p);
    if (std::fabs(det) < 1e-6f) return false;
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

struct ProbeRng {
    uint64_t s;
    float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffffu) * (1.0f / 16777216.0f); }
};

/* node visits of one any-hit walk; FIXED: the preferred child of every pair first, else near first */
template <bool FIXED>
inline uint32_t walk(const rpt_bvh_node *nodes, const rpt_per_vertex_data *pv, const rpt_triangle *idx, const std::vector<uint8_t> &flip, V o, V d,
                     float max_t, bool &occluded) {
    uint32_t stack[64];
    int sp = 0;
    uint32_t visits = 0, node = 0;
    occluded = false;
    const V id{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    for (;;) {
        visits += 1;
        const rpt_bvh_node &n = nodes[node];
        bool descend = false;
        if (n.triangle_count != 0u) {
            for (uint32_t k = 0; k < n.triangle_count; ++k)
                if (tri(pv, idx[n.left_or_first + k], o, d, max_t)) { occluded = true; return visits; }
        } else {
            const uint32_t L = n.left_or_first, R = L + 1u;
            float tl, tr;
            const bool hl = box(nodes[L], o, id, tl), hr = box(nodes[R], o, id, tr);
            const bool right = FIXED ? (hr && (!hl || flip[L >> 1] != 0)) : (hr && (!hl || tl > tr));
            if (hl || hr) {
                if (hl && hr && sp < 64) stack[sp++] = right ? L : R;
                node = right ? R : L;
                descend = true;
            }
        }
        if (!descend) {
            if (sp == 0) return visits;
            node = stack[--sp];
        }
    }
}

}  // namespace shadow_order_detail

/* Expects a validated scene (rpt_hip.hip validate_scene: links in range, no cycles, leaf ranges inside the index buffer).  `pair_shaped`: children of
 * every inner node are the nodes (2p + 1, 2p + 2) of one pair — what the flipped copies can express; otherwise near-first stays. */
inline ShadowOrder choose_shadow_order(const rpt_per_vertex_data *pv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn,
                                       const rpt_material_data *mats, const rpt_light_pick_entry *lp, size_t nlp, bool pair_shaped,
                                       const float *cross_sq = nullptr /* optional: |(b - a) x (c - a)|^2 per triangle, as the upload's derive kernel hands it out */) {
    using namespace shadow_order_detail;
    ShadowOrder so;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Stamp { ShadowOrder &so; std::chrono::steady_clock::time_point t0; ~Stamp() { so.probe_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } stamp{so, t_begin};
    if (nlp == 0 || lp[0].ratio < 0.0f || nt == 0) return so;
    if (!pair_shaped || nn < 3) { so.why = "node pool is not pair-shaped"; return so; }
    /* triangle surface below every node, children before parents (explicit post-order: the pool's index order is the builder's business) */
    std::vector<double> area(nn, 0.0), tri_area(nt, 0.0);
    for (size_t t = 0; t < nt; ++t) {
        if (cross_sq) {                        /* (three gathers of 64-byte vertices per triangle are most of this function on a 1 M-triangle scene) */
            tri_area[t] = 0.5 * std::sqrt((double)cross_sq[t]);
        } else {
            V a = vtx(pv[idx[t].v0]);
            V x = cross(sub(vtx(pv[idx[t].v1]), a), sub(vtx(pv[idx[t].v2]), a));
            tri_area[t] = 0.5 * std::sqrt((double)dot(x, x));
        }
    }
    {
        std::vector<uint32_t> order;
        order.reserve(nn);
        std::vector<uint32_t> st(1, 0u);
        while (!st.empty()) {
            uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            if (nodes[n].triangle_count == 0u) { st.push_back(nodes[n].left_or_first); st.push_back(nodes[n].left_or_first + 1u); }
        }
        for (size_t i = order.size(); i-- > 0;) {
            const rpt_bvh_node &n = nodes[order[i]];
            if (n.triangle_count == 0u) area[order[i]] = area[n.left_or_first] + area[n.left_or_first + 1u];
            else for (uint32_t k = 0; k < n.triangle_count; ++k) area[order[i]] += tri_area[n.left_or_first + k];
        }
    }
    auto opacity = [&](uint32_t n) {
        const rpt_bvh_node &b = nodes[n];
        const double ex = (double)b.aabb_max[0] - b.aabb_min[0], ey = (double)b.aabb_max[1] - b.aabb_min[1], ez = (double)b.aabb_max[2] - b.aabb_min[2];
        const double half = std::max(ex * ey + ey * ez + ez * ex, 1e-30);
        return std::min(1.0, area[n] / half);
    };
    so.flip.assign((nn - 1) / 2, 0);
    for (size_t p = 0; p < so.flip.size(); ++p) so.flip[p] = opacity((uint32_t)(2 * p + 2)) > opacity((uint32_t)(2 * p + 1)) ? 1 : 0;

    /* probe rays: surface point by area on the non-emissive triangles (all triangles if everything emits), light point as pick_light +
     * pick_triangle_point draw it (light_pick.rs:8-23, 100-134) */
    std::vector<double> cdf(nt);
    double total = 0.0;
    auto emissive = [&](size_t t) { const float *e = mats[idx[t].material].emissive; return e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f; };
    for (size_t t = 0; t < nt; ++t) { total += emissive(t) ? 0.0 : tri_area[t]; cdf[t] = total; }
    if (!(total > 0.0)) { total = 0.0; for (size_t t = 0; t < nt; ++t) { total += tri_area[t]; cdf[t] = total; } }
    if (!(total > 0.0)) { so.why = "degenerate geometry"; return so; }
    ProbeRng rng{0x9e3779b97f4a7c15ull};
    auto point_on = [&](size_t t, float r1, float r2) {
        V a = vtx(pv[idx[t].v0]), b = vtx(pv[idx[t].v1]), c = vtx(pv[idx[t].v2]);
        const float s = std::sqrt(r1), wa = 1.0f - s, wb = s * (1.0f - r2), wc = s * r2;
        return V{wa * a.x + wb * b.x + wc * c.x, wa * a.y + wb * b.y + wc * c.y, wa * a.z + wb * b.z + wc * c.z};
    };
    auto mean_normal = [&](size_t t) {
        const float *a = pv[idx[t].v0].normal, *b = pv[idx[t].v1].normal, *c = pv[idx[t].v2].normal;
        return V{a[0] + b[0] + c[0], a[1] + b[1] + c[1], a[2] + b[2] + c[2]};
    };
    uint64_t vn = 0, vf = 0;
    for (int i = 0; i < SHADOW_PROBE_RAYS; ++i) {
        const double pick = (double)rng.next() * total;
        const size_t t0 = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), pick) - cdf.begin());
        const V p = point_on(std::min(t0, nt - 1), rng.next(), rng.next());
        const rpt_light_pick_entry &e = lp[std::min((size_t)(rng.next() * (float)nlp), nlp - 1)];
        const uint32_t lt = rng.next() < e.ratio ? e.triangle_index_a : e.triangle_index_b;
        if (lt >= nt) continue;
        const V q = point_on(lt, rng.next(), rng.next());
        V d = sub(q, p);
        const float dist = std::sqrt(dot(d, d));
        if (!(dist > 1e-4f)) continue;
        d = V{d.x / dist, d.y / dist, d.z / dist};
        /* only rays the device walks: a light point that faces away (light_pdf = 0) or lies below the surface's horizon (bsdf_pdf = 0) adds a zero
         * term whatever its shadow ray finds, and k_shade does not queue that ray (about half of all NEE evaluations) */
        if (!(dot(mean_normal(std::min(t0, nt - 1)), d) > 0.0f) || !(dot(mean_normal(lt), d) < 0.0f)) continue;
        const V o = V{p.x + d.x * 0.001f, p.y + d.y * 0.001f, p.z + d.z * 0.001f};          /* light_pick.rs:141-147, EPS = 0.001 */
        bool occ_n = false, occ_f = false;
        vn += walk<false>(nodes, pv, idx, so.flip, o, d, dist - 0.002f, occ_n);
        vf += walk<true>(nodes, pv, idx, so.flip, o, d, dist - 0.002f, occ_f);
        so.probe_rays += 1;
        so.probe_occluded += occ_n ? 1u : 0u;
    }
    if (so.probe_rays == 0) { so.why = "no probe ray could be formed"; return so; }
    so.visits_near = (double)vn / so.probe_rays;
    so.visits_fixed = (double)vf / so.probe_rays;
    so.fixed = so.visits_fixed < SHADOW_FIXED_GAIN * so.visits_near;
    so.why = so.fixed ? "opaque-first needs fewer node visits on the probe rays" : "near-first needs no more node visits on the probe rays";
    if (const char *env = getenv("RPT_SHADOW_ORDER")) {
        if (!strcmp(env, "fixed")) { so.fixed = true; so.why = "RPT_SHADOW_ORDER=fixed"; }
        else if (!strcmp(env, "near")) { so.fixed = false; so.why = "RPT_SHADOW_ORDER=near"; }
    }
    return so;
}

/* ---- the hit-or-miss lanes of the last extension rays (k_traverse.h k_traverse_nearest_stream LAST) -------------------------------------------------
 * Without NEE the last extension ray of a path that cannot end on an emitter only has to say "hit or miss": the part of the reference's walk up to its first
 * accepted triangle, which is an any-hit walk (result.t is 1e6 throughout) and as free in its order as a shadow query.  These rays are not shadow rays — they
 * leave a surface in a direction the BSDF drew, and in a closed scene all of them hit — so the order is chosen on rays of their kind: points by area on the
 * non-emissive triangles, cosine-distributed directions about the shading normal, walked on the host near child first (the primary image: no second copy
 * needed) and in fixed order under three rules that put into the left slot the child that is (1) more opaque, (2) the smaller subtree, (3) more opaque per
 * node of its subtree — the classic "most likely per unit of cost first" for a search that stops at its first success.  tools/last_bounce_sim.py and the
 * replay of the real bounce-3 rays of DarkCornell: 18.9 node visits near first, 15.2 / 12.2 / 11.9 under the three rules (the whole walk: 25.8).  LDS-image scenes only
 * (the only ones with a LAST kernel): a few hundred nodes, the probe is a fraction of a millisecond. */
#define LAST_PROBE_RAYS 1024
struct LastOrder {
    int rule = 0;                       /* 0: near child first on the primary image; 1..3: fixed order over a copy flipped by that rule */
    std::vector<uint8_t> flip;
    double visits[4] = {0.0, 0.0, 0.0, 0.0};   /* node visits per probe ray: near first, rules 1..3 */
    uint32_t probe_rays = 0, probe_hits = 0;
    double probe_ms = 0.0;
};

inline LastOrder choose_last_order(const rpt_per_vertex_data *pv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn,
                                   const rpt_material_data *mats, bool pair_shaped) {
    using namespace shadow_order_detail;
    LastOrder lo;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Stamp { LastOrder &lo; std::chrono::steady_clock::time_point t0; ~Stamp() { lo.probe_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } stamp{lo, t_begin};
    if (!pair_shaped || nn < 3 || nt == 0) return lo;
    std::vector<double> area(nn, 0.0), tri_area(nt, 0.0), count(nn, 1.0);
    for (size_t t = 0; t < nt; ++t) {
        V a = vtx(pv[idx[t].v0]);
        V x = cross(sub(vtx(pv[idx[t].v1]), a), sub(vtx(pv[idx[t].v2]), a));
        tri_area[t] = 0.5 * std::sqrt((double)dot(x, x));
    }
    {
        std::vector<uint32_t> order;
        order.reserve(nn);
        std::vector<uint32_t> st(1, 0u);
        while (!st.empty()) {
            uint32_t n = st.back(); st.pop_back();
            order.push_back(n);
            if (nodes[n].triangle_count == 0u) { st.push_back(nodes[n].left_or_first); st.push_back(nodes[n].left_or_first + 1u); }
        }
        for (size_t i = order.size(); i-- > 0;) {
            const rpt_bvh_node &n = nodes[order[i]];
            if (n.triangle_count == 0u) {
                area[order[i]] = area[n.left_or_first] + area[n.left_or_first + 1u];
                count[order[i]] = 1.0 + count[n.left_or_first] + count[n.left_or_first + 1u];
            } else for (uint32_t k = 0; k < n.triangle_count; ++k) area[order[i]] += tri_area[n.left_or_first + k];
        }
    }
    auto opacity = [&](uint32_t n) {
        const rpt_bvh_node &b = nodes[n];
        const double ex = (double)b.aabb_max[0] - b.aabb_min[0], ey = (double)b.aabb_max[1] - b.aabb_min[1], ez = (double)b.aabb_max[2] - b.aabb_min[2];
        return std::min(1.0, area[n] / std::max(ex * ey + ey * ez + ez * ex, 1e-30));
    };
    const size_t P = (nn - 1) / 2;
    std::vector<uint8_t> flips[4];
    for (int r = 1; r <= 3; ++r) flips[r].assign(P, 0);
    for (size_t p = 0; p < P; ++p) {
        const uint32_t L = (uint32_t)(2 * p + 1), R = L + 1u;
        flips[1][p] = opacity(R) > opacity(L) ? 1 : 0;
        flips[2][p] = count[R] < count[L] ? 1 : 0;
        flips[3][p] = opacity(R) / count[R] > opacity(L) / count[L] ? 1 : 0;
    }
    std::vector<double> cdf(nt);
    double total = 0.0;
    auto emissive = [&](size_t t) { const float *e = mats[idx[t].material].emissive; return e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f; };
    for (size_t t = 0; t < nt; ++t) { total += emissive(t) ? 0.0 : tri_area[t]; cdf[t] = total; }
    if (!(total > 0.0)) return lo;
    ProbeRng rng{0x2545f4914f6cdd1dull};
    uint64_t v[4] = {0, 0, 0, 0};
    for (int i = 0; i < LAST_PROBE_RAYS; ++i) {
        const double pick = (double)rng.next() * total;
        const size_t t0 = std::min((size_t)(std::lower_bound(cdf.begin(), cdf.end(), pick) - cdf.begin()), nt - 1);
        const V a = vtx(pv[idx[t0].v0]), b = vtx(pv[idx[t0].v1]), c = vtx(pv[idx[t0].v2]);
        const float s = std::sqrt(rng.next()), r2 = rng.next(), wa = 1.0f - s, wb = s * (1.0f - r2), wc = s * r2;
        const V p{wa * a.x + wb * b.x + wc * c.x, wa * a.y + wb * b.y + wc * c.y, wa * a.z + wb * b.z + wc * c.z};
        const float *na = pv[idx[t0].v0].normal, *nb = pv[idx[t0].v1].normal, *nc = pv[idx[t0].v2].normal;
        V n{na[0] + nb[0] + nc[0], na[1] + nb[1] + nc[1], na[2] + nb[2] + nc[2]};
        const float nl = std::sqrt(dot(n, n));
        const float u1 = rng.next(), u2 = rng.next();
        if (!(nl > 1e-12f)) continue;
        n = V{n.x / nl, n.y / nl, n.z / nl};
        /* cosine-distributed direction about n */
        const V h = std::fabs(n.x) < 0.5f ? V{1, 0, 0} : V{0, 1, 0};
        V t = cross(n, h);
        const float tl = std::sqrt(dot(t, t));
        t = V{t.x / tl, t.y / tl, t.z / tl};
        const V bt = cross(n, t);
        const float r = std::sqrt(u1), phi = 6.2831853f * u2, x = r * std::cos(phi), y = r * std::sin(phi), z = std::sqrt(std::fmax(0.0f, 1.0f - u1));
        const V d{x * t.x + y * bt.x + z * n.x, x * t.y + y * bt.y + z * n.y, x * t.z + y * bt.z + z * n.z};
        if (d.x == 0.0f || d.y == 0.0f || d.z == 0.0f) continue;
        const V o{p.x + n.x * 0.001f, p.y + n.y * 0.001f, p.z + n.z * 0.001f};
        bool hit = false;
        v[0] += walk<false>(nodes, pv, idx, flips[1], o, d, 1000000.0f, hit);
        for (int q = 1; q <= 3; ++q) v[q] += walk<true>(nodes, pv, idx, flips[q], o, d, 1000000.0f, hit);
        lo.probe_rays += 1;
        lo.probe_hits += hit ? 1u : 0u;
    }
    if (lo.probe_rays == 0) return lo;
    for (int q = 0; q < 4; ++q) lo.visits[q] = (double)v[q] / lo.probe_rays;
    int best = 1;
    for (int q = 2; q <= 3; ++q) if (lo.visits[q] < lo.visits[best]) best = q;
    lo.rule = lo.visits[best] < SHADOW_FIXED_GAIN * lo.visits[0] ? best : 0;
    if (const char *env = getenv("RPT_LAST_ORDER")) {              /* near | opaque | small | ratio: tests and A/B */
        if (!strcmp(env, "near")) lo.rule = 0; else if (!strcmp(env, "opaque")) lo.rule = 1; else if (!strcmp(env, "small")) lo.rule = 2; else if (!strcmp(env, "ratio")) lo.rule = 3;
    }
    if (lo.rule != 0) lo.flip = flips[lo.rule];
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
