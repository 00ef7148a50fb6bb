/*
 * anyhit_order_sim.cpp — what does the ORDER FREEDOM of the any-hit walk buy?  (CPU analysis tool, no GPU; driven by tools/anyhit_order_sim.py.)
 *
 * The reference's shadow query (kernels/src/light_pick.rs:141-148 -> intersection.rs:173-234 with NEAREST_HIT = false) consumes `.hit` only.  Until the
 * first accepted triangle `result.t` stays 1e6, every box is tested against that constant (:212-213), and the first accept returns (:201-203): the set
 * of boxes a ray MAY enter does not depend on the order its children are visited in, and `.hit` is the OR of the accept test over the triangles of the
 * reachable leaves.  Any visiting order is therefore bit-exact for `.hit` (tests/test_anyhit_order.py proves it on 10^6 rays through the oracle).
 *
 * This replays the streamed any-hit kernels (k_traverse.h: one ray per lane, one body per trip chosen by majority, refill of idle lanes every TRIPS trips
 * once REFILL lanes are idle) on the real shadow rays of a scene — dumped by the oracle's trace_pixel in queue order — under
 *   0 near     the reference's near-first order (what the kernels do today)
 *   1 left     fixed left-first order: no `tl > tr`, no swap
 *   2 area     fixed order, the child with the larger box surface first (static, decided at upload)
 *   3 vote     wave-majority: the lanes that hit both children vote with their near-first preference, everybody follows the majority
 *   4 packet   packet descent: ONE per-wave stack of (node, 64-bit lane mask), every node record a scalar load, lanes outside the mask idle
 *   5 packet+r the same with refills: idle lanes get new rays, pushed as (root, mask of the new lanes)
 *   6 thread   stackless threaded layout: DFS order + skip links, ONE box per trip, no LDS stack
 * and reports, per variant: trips by kind, lanes per trip, wave-uniform share of the inner trips, vector-memory instructions per ray, and wave-instructions
 * per ray from per-body instruction counts (read off the disassembly of the shipped kernels, passed in by the driver).
 *
 * Includes the oracle's translation unit for its box / triangle tests and its trace_pixel: analysis, like tools/traversal_sim.py — never the product.
 */
#include "../oracle/rpt_oracle.cpp"

#include <algorithm>

namespace {

struct SimRay { V3 o, d; float max_t; };

struct SimScene {
    Scene sc;
    std::vector<uint32_t> parent;       /* node -> parent */
    std::vector<double> hits_all, hits_first, n_sub, clear_visits;   /* learned: rays with an acceptable triangle below a node (all such leaves / the one near-first finds) */
    std::vector<uint8_t> learned;       /* bit k: right first under learned rule k */
    std::vector<float> op_l, op_r;      /* opacity of the children of an inner node */
    std::vector<uint8_t> right_first;   /* inner node: static preference for the right child; bit h = heuristic h (see make_sim_scene) */
};

inline float half_area(const rpt_bvh_node &n) {
    float ex = n.aabb_max[0] - n.aabb_min[0], ey = n.aabb_max[1] - n.aabb_min[1], ez = n.aabb_max[2] - n.aabb_min[2];
    return ex * ey + ey * ez + ez * ex;
}

SimScene make_sim_scene(const oracle_scene *s) {
    SimScene ss;
    ss.sc = make_scene(s);
    const Scene &sc = ss.sc;
    ss.parent.assign(sc.n_nodes, 0xffffffffu);
    ss.right_first.assign(sc.n_nodes, 0);
    ss.op_l.assign(sc.n_nodes, 0.0f); ss.op_r.assign(sc.n_nodes, 0.0f);
    ss.hits_all.assign(sc.n_nodes, 0.0); ss.hits_first.assign(sc.n_nodes, 0.0); ss.clear_visits.assign(sc.n_nodes, 0.0); ss.learned.assign(sc.n_nodes, 0);
    /* per subtree: node count (the cost of ruling it out) and the surface of its triangles (how much of its box is opaque) */
    std::vector<double> tri_area(sc.n_nodes, 0.0), n_sub(sc.n_nodes, 1.0);
    for (uint32_t i = (uint32_t)sc.n_nodes; i-- > 0;) {            /* children have larger indices than their parent in the reference's pool */
        const rpt_bvh_node &n = sc.nodes[i];
        if (n.triangle_count == 0) {
            ss.parent[n.left_or_first] = i;
            ss.parent[n.left_or_first + 1] = i;
            tri_area[i] = tri_area[n.left_or_first] + tri_area[n.left_or_first + 1];
            n_sub[i] = 1.0 + n_sub[n.left_or_first] + n_sub[n.left_or_first + 1];
        } else {
            for (uint32_t k = 0; k < n.triangle_count; ++k) {
                rpt_triangle t = sc.indices[n.left_or_first + k];
                V3 a = xyz(sc.per_vertex[t.v0].vertex), b = xyz(sc.per_vertex[t.v1].vertex), c = xyz(sc.per_vertex[t.v2].vertex);
                V3 x = cross(b - a, c - a);
                tri_area[i] += 0.5 * std::sqrt((double)dot(x, x));
            }
        }
    }
    for (uint32_t i = 0; i < sc.n_nodes; ++i) {
        const rpt_bvh_node &n = sc.nodes[i];
        if (n.triangle_count != 0) continue;
        const uint32_t L = n.left_or_first, R = L + 1;
        const double aL = std::max((double)half_area(sc.nodes[L]), 1e-12), aR = std::max((double)half_area(sc.nodes[R]), 1e-12);
        const double oL = std::min(1.0, tri_area[L] / aL), oR = std::min(1.0, tri_area[R] / aR);     /* opacity: triangle surface / box half-surface */
        ss.op_l[i] = (float)oL; ss.op_r[i] = (float)oR;
        uint8_t bits = 0;
        if (aR > aL) bits |= 1;                                   /* 0: the larger box first */
        if (oR > oL) bits |= 2;                                   /* 1: the more opaque child first */
        if (n_sub[R] < n_sub[L]) bits |= 4;                       /* 2: the smaller subtree first */
        if (oR / n_sub[R] > oL / n_sub[L]) bits |= 8;             /* 3: opacity per node of the subtree */
        if (aR * oR / n_sub[R] > aL * oL / n_sub[L]) bits |= 16;  /* 4: opaque surface per node */
        if (tri_area[R] > tri_area[L]) bits |= 32;                /* 5: more triangle surface first */
        ss.right_first[i] = bits;
    }
    ss.n_sub = n_sub;
    return ss;
}

inline bool box_hit(const Scene &sc, uint32_t node, const SimRay &r, float &dist) {
    const rpt_bvh_node &n = sc.nodes[node];
    dist = intersect_aabb(xyz(n.aabb_min), xyz(n.aabb_max), r.o, r.d, 1000000.0f);      /* any-hit: result.t is 1e6 until the walk returns */
    return !rptm::isinfr(dist);
}

/* the same slab test, also handing out tmax (ordering rules only; the hit decision is intersect_aabb's) */
inline void box_interval(const Scene &sc, uint32_t node, const SimRay &r, float &tmin, float &tmax) {
    const rpt_bvh_node &n = sc.nodes[node];
    float tx1 = (n.aabb_min[0] - r.o.x) / r.d.x, tx2 = (n.aabb_max[0] - r.o.x) / r.d.x;
    tmin = m_min(tx1, tx2); tmax = m_max(tx1, tx2);
    float ty1 = (n.aabb_min[1] - r.o.y) / r.d.y, ty2 = (n.aabb_max[1] - r.o.y) / r.d.y;
    tmin = m_max(tmin, m_min(ty1, ty2)); tmax = m_min(tmax, m_max(ty1, ty2));
    float tz1 = (n.aabb_min[2] - r.o.z) / r.d.z, tz2 = (n.aabb_max[2] - r.o.z) / r.d.z;
    tmin = m_max(tmin, m_min(tz1, tz2)); tmax = m_min(tmax, m_max(tz1, tz2));
}

inline bool tri_accept(const Scene &sc, uint32_t ti, const SimRay &r) {
    rpt_triangle tri = sc.indices[ti];
    V3 a = xyz(sc.per_vertex[tri.v0].vertex), b = xyz(sc.per_vertex[tri.v1].vertex), c = xyz(sc.per_vertex[tri.v2].vertex);
    float t = 0.0f;
    bool bf = false;
    return muller_trumbore(r.o, r.d, a, b, c, t, bf) && t > 0.001f && t < 1000000.0f && t <= r.max_t;
}

constexpr int W = 64;
constexpr uint32_t DEAD = 0xffffffffu;

struct SimOut {
    uint64_t rays, occluded;
    uint64_t inner_trips, inner_uniform, inner_lanes;       /* trips of the inner body, of them wave-uniform, lanes taking part */
    uint64_t leaf_trips, leaf_iters, leaf_lanes;            /* trips of the leaf body, triangle iterations the wave issues, lane-tests */
    uint64_t pop_trips;                                     /* inner / leaf trips in which at least one lane pops (one more load, per-lane stacks) */
    uint64_t refills, skipped;                              /* refill passes; packet: popped entries whose lanes were all finished */
    uint64_t box_tests, tri_tests;                          /* per-lane totals (order-independent for unoccluded rays) */
    uint64_t max_stack;
    uint64_t visits_occluded, visits_clear, rays_clear;     /* node visits of occluded / unoccluded rays */
};

struct Params { int order, trips, refill, heuristic; };

/* ---- per-lane stacks: near / left / area / vote ---------------------------------------------------------------------------------------- */
void sim_lanes(const SimScene &ss, const SimRay *rays, uint32_t n, const Params &p, SimOut &o, uint8_t *hit_out) {
    const Scene &sc = ss.sc;
    uint32_t cur[W], sp[W], st[W][40], ray[W], visits[W];
    bool have[W];
    for (int l = 0; l < W; ++l) { cur[l] = DEAD; sp[l] = 0; have[l] = false; ray[l] = 0; visits[l] = 0; }
    uint32_t next = 0;
    auto finish = [&](int l, bool hit) {
        hit_out[ray[l]] = hit ? 1 : 0;
        if (hit) { o.occluded++; o.visits_occluded += visits[l]; } else { o.visits_clear += visits[l]; o.rays_clear++; }
        cur[l] = DEAD; have[l] = false;
    };
    for (;;) {
        int n_idle = 0;
        for (int l = 0; l < W; ++l) n_idle += cur[l] == DEAD;
        const bool more = next < n;
        if ((more && n_idle >= p.refill) || n_idle == W) {
            if (!more && n_idle == W) break;
            for (int l = 0; l < W && next < n; ++l)
                if (cur[l] == DEAD) { ray[l] = next++; cur[l] = 0; sp[l] = 0; have[l] = true; visits[l] = 0; o.rays++; }
            o.refills++;
            continue;
        }
        const int budget = more ? p.trips : 0x7fffffff;
        for (int trip = 0; trip < budget; ++trip) {
            int n_inner = 0, n_leaf = 0;
            for (int l = 0; l < W; ++l)
                if (cur[l] != DEAD) { if (sc.nodes[cur[l]].triangle_count == 0) n_inner++; else n_leaf++; }
            if (n_inner + n_leaf == 0) break;
            const bool do_leaf = n_leaf > n_inner;
            bool any_pop = false;
            if (!do_leaf) {
                /* vote first (order 3): near-first preference of the lanes that hit both children */
                bool hl[W], hr[W], nearR[W];
                int votes_r = 0, votes_l = 0;
                uint32_t first_node = DEAD;
                bool uniform = true;
                for (int l = 0; l < W; ++l) {
                    if (cur[l] == DEAD || sc.nodes[cur[l]].triangle_count != 0) continue;
                    if (first_node == DEAD) first_node = cur[l]; else if (cur[l] != first_node) uniform = false;
                    const uint32_t L = sc.nodes[cur[l]].left_or_first;
                    float dl, dr;
                    hl[l] = box_hit(sc, L, rays[ray[l]], dl);
                    hr[l] = box_hit(sc, L + 1, rays[ray[l]], dr);
                    nearR[l] = hr[l] && (!hl[l] || dl > dr);
                    if (hl[l] && hr[l]) { if (nearR[l]) votes_r++; else votes_l++; }
                    o.box_tests += 2;
                }
                o.inner_trips++; o.inner_lanes += (uint64_t)n_inner; o.inner_uniform += uniform ? 1 : 0;
                for (int l = 0; l < W; ++l) {
                    if (cur[l] == DEAD || sc.nodes[cur[l]].triangle_count != 0) continue;
                    visits[l]++;
                    const uint32_t node = cur[l], L = sc.nodes[node].left_or_first;
                    bool right;
                    switch (p.order) {
                        case 0: right = nearR[l]; break;
                        case 1: right = hr[l] && !hl[l]; break;
                        case 2: right = hr[l] && (!hl[l] || ((ss.right_first[node] >> p.heuristic) & 1)); break;
                        case 3: right = hr[l] && (!hl[l] || votes_r > votes_l); break;
                        case 8: right = hr[l] && (!hl[l] || ((ss.learned[node] >> p.heuristic) & 1)); break;
                        default: {
                            /* dynamic rules on what the slab tests computed anyway: (tmin, tmax) of both children and the ray's max_t */
                            right = hr[l] && !hl[l];
                            if (hl[l] && hr[l]) {
                                const SimRay &r = rays[ray[l]];
                                float nl, fl, nr, fr;
                                box_interval(sc, L, r, nl, fl);
                                box_interval(sc, L + 1, r, nr, fr);
                                const bool opaqueR = (ss.right_first[node] >> 1) & 1;
                                const bool beyondL = nl > r.max_t, beyondR = nr > r.max_t;             /* nothing in there can be accepted (up to rounding): last */
                                const bool throughL = nl > 0.0f && fl <= r.max_t, throughR = nr > 0.0f && fr <= r.max_t;   /* the segment crosses the whole box */
                                const bool insideL = nl <= 0.0f, insideR = nr <= 0.0f;                   /* the origin is in the box: the ray's own surface */
                                switch (p.heuristic) {
                                    case 0: right = beyondL != beyondR ? beyondL : nearR[l]; break;                                  /* in range first, then near */
                                    case 1: right = beyondL != beyondR ? beyondL : opaqueR; break;                                   /* in range first, then opaque */
                                    case 2: right = throughL != throughR ? throughR : nearR[l]; break;                               /* crossed boxes first, then near */
                                    case 3: right = throughL != throughR ? throughR : opaqueR; break;                                /* crossed boxes first, then opaque */
                                    case 4: right = insideL != insideR ? insideL : nearR[l]; break;                                  /* boxes holding the origin last, then near */
                                    case 5: right = insideL != insideR ? insideL : opaqueR; break;                                   /* boxes holding the origin last, then opaque */
                                    case 6: right = beyondL != beyondR ? beyondL : (insideL != insideR ? insideL : nearR[l]); break; /* beyond last, origin boxes next to last, near */
                                    case 7: right = beyondL != beyondR ? beyondL : (insideL != insideR ? insideL : opaqueR); break;
                                    case 8: case 9: case 10: case 11: {                                                                   /* opaque first where the opacities differ by a factor, near otherwise */
                                        const float k = p.heuristic == 8 ? 1.25f : (p.heuristic == 9 ? 2.0f : (p.heuristic == 10 ? 4.0f : 8.0f));
                                        const float a = ss.op_l[node], b = ss.op_r[node];
                                        right = b > k * a ? true : (a > k * b ? false : nearR[l]);
                                        break;
                                    }
                                    default: right = nearR[l]; break;
                                }
                            }
                        }
                    }
                    if (hl[l] || hr[l]) {
                        if (hl[l] && hr[l]) { st[l][sp[l]++] = right ? L : L + 1; o.max_stack = std::max<uint64_t>(o.max_stack, sp[l]); }
                        cur[l] = right ? L + 1 : L;
                    } else if (sp[l] == 0) {
                        finish(l, false);
                    } else {
                        cur[l] = st[l][--sp[l]];
                        any_pop = true;
                    }
                }
            } else {
                uint32_t iters = 0;
                o.leaf_trips++; o.leaf_lanes += (uint64_t)n_leaf;
                for (int l = 0; l < W; ++l) {
                    if (cur[l] == DEAD || sc.nodes[cur[l]].triangle_count == 0) continue;
                    visits[l]++;
                    const rpt_bvh_node &nd = sc.nodes[cur[l]];
                    bool acc = false;
                    uint32_t done = 0;
                    for (uint32_t i = 0; i < nd.triangle_count; ++i) {
                        done++; o.tri_tests++;
                        if (tri_accept(sc, nd.left_or_first + i, rays[ray[l]])) { acc = true; break; }
                    }
                    iters = std::max(iters, done);
                    if (acc) finish(l, true);
                    else if (sp[l] == 0) finish(l, false);
                    else { cur[l] = st[l][--sp[l]]; any_pop = true; }
                }
                o.leaf_iters += iters;
            }
            o.pop_trips += any_pop ? 1 : 0;
        }
    }
}

/* ---- packet descent: one stack of (node, lane mask) per wave ---------------------------------------------------------------------------- */
void sim_packet(const SimScene &ss, const SimRay *rays, uint32_t n, const Params &p, bool refill, SimOut &o, uint8_t *hit_out) {
    const Scene &sc = ss.sc;
    struct Ent { uint32_t node; uint64_t mask; };
    std::vector<Ent> stack;
    uint32_t ray[W], visits[W];
    uint64_t alive = 0;                 /* lanes whose walk has not ended by an accept */
    uint64_t holding = 0;               /* lanes that hold a ray whose result is not noted yet */
    bool hitf[W];
    for (int l = 0; l < W; ++l) { ray[l] = 0; visits[l] = 0; hitf[l] = false; }
    uint32_t next = 0;
    auto note = [&](int l) {
        hit_out[ray[l]] = hitf[l] ? 1 : 0;
        if (hitf[l]) { o.occluded++; o.visits_occluded += visits[l]; } else { o.visits_clear += visits[l]; o.rays_clear++; }
        holding &= ~(1ull << l);
    };
    auto live_mask = [&]() { uint64_t m = 0; for (const Ent &e : stack) m |= e.mask; return m & alive; };
    auto deal = [&](uint64_t idle) {
        uint64_t fresh = 0;
        for (int l = 0; l < W && next < n; ++l)
            if (idle >> l & 1) {
                if (holding >> l & 1) note(l);
                ray[l] = next++; visits[l] = 0; hitf[l] = false; fresh |= 1ull << l; o.rays++;
            }
        alive |= fresh; holding |= fresh;
        if (fresh) stack.push_back(Ent{0u, fresh});
        o.refills++;
    };
    int since = 0;
    for (;;) {
        if (stack.empty()) {
            if (next >= n) break;
            deal(~0ull);
            continue;
        }
        if (refill && next < n && ++since >= p.trips) {
            since = 0;
            const uint64_t idle = ~live_mask();
            if (__builtin_popcountll(idle) >= p.refill) { deal(idle); continue; }
        }
        Ent e = stack.back();
        stack.pop_back();
        const uint64_t m = e.mask & alive;
        if (m == 0) { o.skipped++; continue; }
        const rpt_bvh_node &nd = sc.nodes[e.node];
        const int lanes = __builtin_popcountll(m);
        if (nd.triangle_count == 0) {
            uint64_t mL = 0, mR = 0;
            for (int l = 0; l < W; ++l)
                if (m >> l & 1) {
                    float d;
                    visits[l]++;
                    if (box_hit(sc, nd.left_or_first, rays[ray[l]], d)) mL |= 1ull << l;
                    if (box_hit(sc, nd.left_or_first + 1, rays[ray[l]], d)) mR |= 1ull << l;
                    o.box_tests += 2;
                }
            o.inner_trips++; o.inner_uniform++; o.inner_lanes += (uint64_t)lanes;
            /* the child with more lanes is walked first (its lanes that find an occluder leave the other child's mask sooner) */
            const bool right_first = __builtin_popcountll(mR) > __builtin_popcountll(mL);
            if (right_first) { if (mL) stack.push_back(Ent{nd.left_or_first, mL}); if (mR) stack.push_back(Ent{nd.left_or_first + 1, mR}); }
            else { if (mR) stack.push_back(Ent{nd.left_or_first + 1, mR}); if (mL) stack.push_back(Ent{nd.left_or_first, mL}); }
            o.max_stack = std::max<uint64_t>(o.max_stack, stack.size());
        } else {
            uint64_t left = m;
            uint32_t iters = 0;
            o.leaf_trips++; o.leaf_lanes += (uint64_t)lanes;
            for (int l = 0; l < W; ++l) if (m >> l & 1) visits[l]++;
            for (uint32_t i = 0; i < nd.triangle_count && left; ++i) {
                iters++;
                for (int l = 0; l < W; ++l)
                    if (left >> l & 1) {
                        o.tri_tests++;
                        if (tri_accept(sc, nd.left_or_first + i, rays[ray[l]])) { hitf[l] = true; left &= ~(1ull << l); alive &= ~(1ull << l); }
                    }
            }
            o.leaf_iters += iters;
        }
    }
    for (int l = 0; l < W; ++l) if (holding >> l & 1) note(l);
}

/* ---- stackless threaded layout: one box per trip, DFS order, skip links ---------------------------------------------------------------- */
void sim_threaded(const SimScene &ss, const SimRay *rays, uint32_t n, const Params &p, SimOut &o, uint8_t *hit_out) {
    const Scene &sc = ss.sc;
    auto skip = [&](uint32_t node) -> uint32_t {
        for (;;) {
            if (node == 0u) return DEAD;
            if (node & 1u) return node + 1u;            /* a left child: its sibling (children are pairs 2p + 1, 2p + 2) */
            node = ss.parent[node];
        }
    };
    uint32_t cur[W], ray[W], visits[W];
    bool in_leaf[W];
    for (int l = 0; l < W; ++l) { cur[l] = DEAD; in_leaf[l] = false; ray[l] = 0; visits[l] = 0; }
    uint32_t next = 0;
    auto finish = [&](int l, bool hit) {
        hit_out[ray[l]] = hit ? 1 : 0;
        if (hit) { o.occluded++; o.visits_occluded += visits[l]; } else { o.visits_clear += visits[l]; o.rays_clear++; }
        cur[l] = DEAD; in_leaf[l] = false;
    };
    for (;;) {
        int n_idle = 0;
        for (int l = 0; l < W; ++l) n_idle += cur[l] == DEAD;
        const bool more = next < n;
        if ((more && n_idle >= p.refill) || n_idle == W) {
            if (!more && n_idle == W) break;
            for (int l = 0; l < W && next < n; ++l)
                if (cur[l] == DEAD) {
                    ray[l] = next++; visits[l] = 0; o.rays++;
                    cur[l] = sc.nodes[0].triangle_count == 0 ? sc.nodes[0].left_or_first : 0u;     /* the root's own box is never tested (intersection.rs:180) */
                    in_leaf[l] = sc.nodes[0].triangle_count != 0;
                }
            o.refills++;
            continue;
        }
        const int budget = more ? p.trips : 0x7fffffff;
        for (int trip = 0; trip < budget; ++trip) {
            int n_box = 0, n_leaf = 0;
            for (int l = 0; l < W; ++l)
                if (cur[l] != DEAD) { if (in_leaf[l]) n_leaf++; else n_box++; }
            if (n_box + n_leaf == 0) break;
            if (n_leaf <= n_box) {
                uint32_t first_node = DEAD;
                bool uniform = true;
                o.inner_trips++; o.inner_lanes += (uint64_t)n_box;
                for (int l = 0; l < W; ++l) {
                    if (cur[l] == DEAD || in_leaf[l]) continue;
                    if (first_node == DEAD) first_node = cur[l]; else if (cur[l] != first_node) uniform = false;
                    float d;
                    o.box_tests++;
                    visits[l]++;
                    if (box_hit(sc, cur[l], rays[ray[l]], d)) {
                        if (sc.nodes[cur[l]].triangle_count == 0) cur[l] = sc.nodes[cur[l]].left_or_first;
                        else in_leaf[l] = true;
                    } else {
                        cur[l] = skip(cur[l]);
                        if (cur[l] == DEAD) finish(l, false);
                    }
                }
                o.inner_uniform += uniform ? 1 : 0;
            } else {
                uint32_t iters = 0;
                o.leaf_trips++; o.leaf_lanes += (uint64_t)n_leaf;
                for (int l = 0; l < W; ++l) {
                    if (cur[l] == DEAD || !in_leaf[l]) continue;
                    const rpt_bvh_node &nd = sc.nodes[cur[l]];
                    bool acc = false;
                    uint32_t done = 0;
                    for (uint32_t i = 0; i < nd.triangle_count; ++i) {
                        done++; o.tri_tests++;
                        if (tri_accept(sc, nd.left_or_first + i, rays[ray[l]])) { acc = true; break; }
                    }
                    iters = std::max(iters, done);
                    in_leaf[l] = false;
                    if (acc) finish(l, true);
                    else { cur[l] = skip(cur[l]); if (cur[l] == DEAD) finish(l, false); }
                }
                o.leaf_iters += iters;
            }
        }
    }
}

}  // namespace

extern "C" {

/* shadow rays of the given pixels (x | y << 16) for sample rng[i] at bounce `bounce`: 8 floats each (origin, direction, max_t, light index), valid[i] = 0 when
 * that sample traces no shadow ray at that bounce */
int sim_dump_shadow_rays(const rpt_tracing_config *config, const oracle_scene *scene, const rpt_rng_state *rng_full, uint32_t sample, uint32_t bounce,
                         const uint32_t *pixels, size_t n_pixels, float *shadow, uint8_t *valid) {
    Scene sc = make_scene(scene);
    Counters cnt;
    g_ray_dump_bounce = bounce;
    uint64_t dead_counters[4] = {0, 0, 0, 0};
    g_dead_shadow_rays = dead_counters;             /* rays whose term is zero whatever the walk finds are marked (light index -1): the device does not walk them */
    for (size_t i = 0; i < n_pixels; ++i) {
        const uint32_t x = pixels[i] & 0xffffu, y = pixels[i] >> 16;
        rpt_rng_state r = rng_full[(size_t)y * config->width + x];
        r.n += sample;
        g_shadow_dump = shadow + 8 * i;
        g_shadow_dump_hit = false;
        trace_pixel(x, y, *config, r, sc, cnt);
        valid[i] = g_shadow_dump_hit ? (shadow[8 * i + 7] < 0.0f ? 2 : 1) : 0;
    }
    g_shadow_dump = nullptr;
    g_dead_shadow_rays = nullptr;
    return 0;
}

/* one wave walks rays[0 .. n) (8 floats each, queue order); order as in the header; out = SimOut as 24 uint64; hit_out[n] */
static thread_local const oracle_scene *cached = nullptr;
static thread_local SimScene ss;

/* training pass for the learned static orders: per node, how many of these rays have an acceptable triangle below it */
int sim_learn(const oracle_scene *scene, const float *rays8, uint32_t n, int finish) {
    if (cached != scene) { ss = make_sim_scene(scene); cached = scene; }
    const Scene &sc = ss.sc;
    Counters cnt;
    std::vector<uint32_t> st;
    for (uint32_t i = 0; i < n; ++i) {
        SimRay r{xyz(rays8 + 8 * i), xyz(rays8 + 8 * i + 3), rays8[8 * i + 6]};
        /* every leaf with an acceptable triangle (a full walk, no early exit) */
        st.assign(1, 0u);
        bool any = false;
        std::vector<uint32_t> visited;
        while (!st.empty()) {
            uint32_t node = st.back(); st.pop_back();
            visited.push_back(node);
            const rpt_bvh_node &nd = sc.nodes[node];
            if (nd.triangle_count > 0) {
                bool acc = false;
                for (uint32_t k = 0; k < nd.triangle_count && !acc; ++k) acc = tri_accept(sc, nd.left_or_first + k, r);
                if (acc) { any = true; for (uint32_t a = node; a != 0xffffffffu; a = ss.parent[a]) ss.hits_all[a] += 1.0; }
            } else {
                float d;
                if (box_hit(sc, nd.left_or_first + 1, r, d)) st.push_back(nd.left_or_first + 1);
                if (box_hit(sc, nd.left_or_first, r, d)) st.push_back(nd.left_or_first);
            }
        }
        if (!any) for (uint32_t v : visited) ss.clear_visits[v] += 1.0;
        /* the one the reference's near-first walk finds */
        TraceResult tr = intersect_front_to_back<false>(sc, r.o, r.d, r.max_t, cnt);
        if (tr.hit) {
            uint32_t leaf = 0xffffffffu;
            for (uint32_t v = 0; v < sc.n_nodes; ++v)       /* (tool code: linear search for the leaf that holds the triangle) */
                if (sc.nodes[v].triangle_count > 0 && tr.triangle_index >= sc.nodes[v].left_or_first && tr.triangle_index < sc.nodes[v].left_or_first + sc.nodes[v].triangle_count) { leaf = v; break; }
            for (uint32_t a = leaf; a != 0xffffffffu; a = ss.parent[a]) ss.hits_first[a] += 1.0;
        }
    }
    if (finish) {
        for (uint32_t i = 0; i < sc.n_nodes; ++i) {
            const rpt_bvh_node &nd = sc.nodes[i];
            if (nd.triangle_count != 0) continue;
            const uint32_t L = nd.left_or_first, R = L + 1;
            uint8_t bits = 0;
            auto rule = [&](const std::vector<double> &h, double a) { return h[R] / std::pow(ss.n_sub[R], a) > h[L] / std::pow(ss.n_sub[L], a); };
            if (rule(ss.hits_all, 0.0)) bits |= 1;
            if (rule(ss.hits_all, 0.5)) bits |= 2;
            if (rule(ss.hits_all, 1.0)) bits |= 4;
            if (rule(ss.hits_first, 0.0)) bits |= 8;
            if (rule(ss.hits_first, 0.5)) bits |= 16;
            if (rule(ss.hits_first, 1.0)) bits |= 32;
            ss.learned[i] = bits;
        }
    }
    return 0;
}

int sim_wave(const oracle_scene *scene, const float *rays8, uint32_t n, int order, int trips, int refill, uint64_t *out, uint8_t *hit_out) {
    if (cached != scene) { ss = make_sim_scene(scene); cached = scene; }
    std::vector<SimRay> rays(n);
    for (uint32_t i = 0; i < n; ++i) rays[i] = SimRay{xyz(rays8 + 8 * i), xyz(rays8 + 8 * i + 3), rays8[8 * i + 6]};
    SimOut o{};
    Params p{order >= 50 ? 8 : (order >= 30 ? 7 : (order >= 20 ? 2 : order)), trips, refill, order >= 50 ? order - 50 : (order >= 30 ? order - 30 : (order >= 20 ? order - 20 : 0))};
    order = p.order;
    if (order <= 3 || order == 7 || order == 8) sim_lanes(ss, rays.data(), n, p, o, hit_out);
    else if (order == 4) sim_packet(ss, rays.data(), n, p, false, o, hit_out);
    else if (order == 5) sim_packet(ss, rays.data(), n, p, true, o, hit_out);
    else sim_threaded(ss, rays.data(), n, p, o, hit_out);
    memcpy(out, &o, sizeof(o));
    return (int)(sizeof(o) / sizeof(uint64_t));
}

/* `.hit` of the any-hit query under an arbitrary visiting order (tests/test_anyhit_order.py): mode 0 the reference's walk itself (intersect_front_to_back<false>),
 * 1 left first, 2 right first, 3 far first, 4 a pseudo-random choice per (ray, node), 5 breadth-first over a queue instead of a stack */
int sim_any_hit_order(const oracle_scene *scene, size_t n, const float *origins, const float *dirs, const float *max_t, int mode, uint32_t seed, uint8_t *hit_out) {
    Scene sc = make_scene(scene);
    Counters cnt;
    std::vector<uint32_t> work;
    for (size_t i = 0; i < n; ++i) {
        SimRay r{xyz(origins + 3 * i), xyz(dirs + 3 * i), max_t[i]};
        if (mode == 0) { hit_out[i] = intersect_front_to_back<false>(sc, r.o, r.d, r.max_t, cnt).hit ? 1 : 0; continue; }
        work.clear();
        work.push_back(0u);
        size_t head = 0;
        bool hit = false;
        while (!hit && head < work.size()) {
            uint32_t node;
            if (mode == 5) node = work[head++]; else { node = work.back(); work.pop_back(); }
            const rpt_bvh_node &nd = sc.nodes[node];
            if (nd.triangle_count > 0) {
                for (uint32_t k = 0; k < nd.triangle_count && !hit; ++k) hit = tri_accept(sc, nd.left_or_first + k, r);
            } else {
                float dl, dr;
                const bool hl = box_hit(sc, nd.left_or_first, r, dl), hr = box_hit(sc, nd.left_or_first + 1, r, dr);
                bool right_first;
                switch (mode) {
                    case 1: right_first = false; break;
                    case 2: right_first = true; break;
                    case 3: right_first = !(dl > dr); break;                       /* far first */
                    default: { uint32_t h = (uint32_t)i * 2654435761u ^ node * 0x9e3779b9u ^ seed; h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; right_first = (h & 1u) != 0u; }
                }
                /* a stack pops the LAST push first */
                const uint32_t a = right_first ? nd.left_or_first + 1 : nd.left_or_first, b = right_first ? nd.left_or_first : nd.left_or_first + 1;
                const bool ha = right_first ? hr : hl, hb = right_first ? hl : hr;
                if (mode == 5) { if (ha) work.push_back(a); if (hb) work.push_back(b); }
                else { if (hb) work.push_back(b); if (ha) work.push_back(a); }
            }
        }
        hit_out[i] = hit ? 1 : 0;
    }
    return 0;
}

}  // extern "C"
