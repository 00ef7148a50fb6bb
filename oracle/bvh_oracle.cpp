/*
 * bvh_oracle.cpp — CPU restatement of the reference's BVH builder.  TEST INFRASTRUCTURE ONLY (like rpt_oracle.cpp: only tests/
 * may load it, as the checker).  It exists so that the device builder (csrc/k_bvh_build.h) and the product's host builder
 * (csrc/host/bvh_build.cpp) are compared with something that is NOT the product: this file follows src/bvh.rs statement
 * by statement — the BVHNode itself as the box type, the Segment array, the sweep, the explicit stack — and shares no code
 * with either of them.
 *
 * Reference lines restated (all under /root/reference):
 *   shared_structs/src/lib.rs:121-191   BVHNode (default = inverted infinite box, counts / indices as u32 bits in .w)
 *   src/bvh.rs:9-33                     BVHNodeExtensions: encapsulate, encapsulate_node, area
 *   src/bvh.rs:59-78                    BVHBuilder::new — centroids (v0 + v1 + v2) / 3.0, node pool of 2N - 1
 *   src/bvh.rs:85-103                   update_node_aabb
 *   src/bvh.rs:178-255                  find_best_split_segmented
 *   src/bvh.rs:257-324                  build
 * glam 0.22 semantics used: Vec3::min / max = component-wise f32::min / f32::max (a NaN operand is ignored);
 * Vec3 / f32 divides every component by the scalar; Rust `f32 as usize` saturates and maps NaN to 0.
 *
 * PARITY PINNING: as for rpt_oracle.cpp — the reference cannot be compiled here; this restatement is pinned by the BVH
 * invariants of SURVEY.md 8c (node count <= 2N - 1, adjacent children, leaf ranges partition [0, N), every triangle inside
 * its leaf box) and by traversal == brute force on the resulting trees (tests/test_oracle_kats.py, tests/test_host.py).
 */
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/rpt/shared_structs.h"

namespace {

const float F_INF = std::numeric_limits<float>::infinity();

struct Vec3 {
    float x, y, z;
    float operator[](size_t axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
};
/* f32::min / f32::max: if one operand is NaN the other is returned */
float rs_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
float rs_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
Vec3 v_min(Vec3 a, Vec3 b) { return Vec3{rs_min(a.x, b.x), rs_min(a.y, b.y), rs_min(a.z, b.z)}; }
Vec3 v_max(Vec3 a, Vec3 b) { return Vec3{rs_max(a.x, b.x), rs_max(a.y, b.y), rs_max(a.z, b.z)}; }
Vec3 v_add(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }

/* shared_structs BVHNode: aabb_min.xyz | triangle_count bits, aabb_max.xyz | left / first index bits */
struct Node {
    float mn[4], mx[4];
    Node() { mn[0] = mn[1] = mn[2] = F_INF; mn[3] = 0.0f; mx[0] = mx[1] = mx[2] = -F_INF; mx[3] = 0.0f; }           /* Default (lib.rs:128-135) */
    uint32_t triangle_count() const { uint32_t u; memcpy(&u, &mn[3], 4); return u; }
    uint32_t first_triangle_index() const { uint32_t u; memcpy(&u, &mx[3], 4); return u; }
    void set_triangle_count(uint32_t c) { memcpy(&mn[3], &c, 4); }
    void set_index(uint32_t i) { memcpy(&mx[3], &i, 4); }                                                       /* left_node_index / first_triangle_index */
    Vec3 aabb_min() const { return Vec3{mn[0], mn[1], mn[2]}; }
    Vec3 aabb_max() const { return Vec3{mx[0], mx[1], mx[2]}; }
    void set_aabb_min(Vec3 v) { mn[0] = v.x; mn[1] = v.y; mn[2] = v.z; }
    void set_aabb_max(Vec3 v) { mx[0] = v.x; mx[1] = v.y; mx[2] = v.z; }
    void encapsulate(Vec3 p) { set_aabb_min(v_min(aabb_min(), p)); set_aabb_max(v_max(aabb_max(), p)); }         /* bvh.rs:16-19 */
    void encapsulate_node(const Node &n) {                                                                        /* bvh.rs:21-27 */
        if (n.aabb_min().x == F_INF) return;
        set_aabb_min(v_min(aabb_min(), n.aabb_min()));
        set_aabb_max(v_max(aabb_max(), n.aabb_max()));
    }
    float area() const {                                                                                          /* bvh.rs:29-32 */
        const Vec3 a = aabb_max(), b = aabb_min();
        const Vec3 e{a.x - b.x, a.y - b.y, a.z - b.z};
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
};
static_assert(sizeof(Node) == 32, "BVHNode is 32 bytes");

size_t as_usize(float v) {            /* Rust `as usize` */
    if (!(v > 0.0f)) return 0;        /* NaN, negatives, zero */
    if (v >= 18446744073709551616.0f) return ~(size_t)0;
    return (size_t)v;
}

struct Builder {
    size_t sah_samples;
    const float *vertices;            /* Vec4 per vertex */
    rpt_triangle *indices;
    size_t n_indices;
    std::vector<Vec3> centroids;
    std::vector<Node> nodes;

    Vec3 vertex(uint32_t i) const { return Vec3{vertices[4 * (size_t)i], vertices[4 * (size_t)i + 1], vertices[4 * (size_t)i + 2]}; }

    void update_node_aabb(size_t node_idx) {                                                                     /* bvh.rs:85-103 */
        Node &node = nodes[node_idx];
        Vec3 aabb_min{F_INF, F_INF, F_INF}, aabb_max{-F_INF, -F_INF, -F_INF};
        for (uint32_t i = 0; i < node.triangle_count(); ++i) {
            const rpt_triangle index = indices[node.first_triangle_index() + i];
            const Vec3 v0 = vertex(index.v0), v1 = vertex(index.v1), v2 = vertex(index.v2);
            aabb_min = v_min(aabb_min, v_min(v_min(v0, v1), v2));
            aabb_max = v_max(aabb_max, v_max(v_max(v0, v1), v2));
        }
        node.set_aabb_min(aabb_min);
        node.set_aabb_max(aabb_max);
    }

    void find_best_split_segmented(const Node &node, size_t &best_axis, float &best_split, float &best_cost) const {   /* bvh.rs:178-255 */
        best_axis = 0;
        best_split = 0.0f;
        best_cost = F_INF;
        struct Segment { Node aabb; uint32_t triangle_count = 0; };
        for (size_t axis = 0; axis < 3; ++axis) {
            float bounds_min = F_INF, bounds_max = -F_INF;
            for (uint32_t i = 0; i < node.triangle_count(); ++i) {
                const Vec3 centroid = centroids[node.first_triangle_index() + i];
                bounds_min = rs_min(bounds_min, centroid[axis]);
                bounds_max = rs_max(bounds_max, centroid[axis]);
            }
            if (bounds_min == bounds_max) continue;                                                              /* completely flat */
            std::vector<Segment> segments(sah_samples);
            float scale = (float)sah_samples / (bounds_max - bounds_min);
            for (uint32_t i = 0; i < node.triangle_count(); ++i) {
                const size_t triangle_index = node.first_triangle_index() + i;
                const rpt_triangle index = indices[triangle_index];
                const Vec3 v0 = vertex(index.v0), v1 = vertex(index.v1), v2 = vertex(index.v2);
                size_t segment_index = as_usize((centroids[triangle_index][axis] - bounds_min) * scale);
                if (segment_index > sah_samples - 1) segment_index = sah_samples - 1;
                segments[segment_index].aabb.encapsulate(v0);
                segments[segment_index].aabb.encapsulate(v1);
                segments[segment_index].aabb.encapsulate(v2);
                segments[segment_index].triangle_count += 1;
            }
            Node left_box, right_box;
            uint32_t left_sum = 0, right_sum = 0;
            std::vector<float> left_areas(sah_samples - 1, 0.0f), right_areas(sah_samples - 1, 0.0f);
            std::vector<uint32_t> left_tri_counts(sah_samples - 1, 0u), right_tri_counts(sah_samples - 1, 0u);
            for (size_t i = 0; i + 1 < sah_samples; ++i) {
                left_sum += segments[i].triangle_count;
                left_tri_counts[i] = left_sum;
                left_box.encapsulate_node(segments[i].aabb);
                left_areas[i] = left_box.area();
                right_sum += segments[sah_samples - 1 - i].triangle_count;
                right_tri_counts[sah_samples - 2 - i] = right_sum;
                right_box.encapsulate_node(segments[sah_samples - 1 - i].aabb);
                right_areas[sah_samples - 2 - i] = right_box.area();
            }
            scale = (bounds_max - bounds_min) / (float)sah_samples;
            for (size_t i = 0; i + 1 < sah_samples; ++i) {
                const float cost = (float)left_tri_counts[i] * left_areas[i] + (float)right_tri_counts[i] * right_areas[i];
                if (cost < best_cost) {
                    best_axis = axis;
                    best_split = bounds_min + scale * (float)(i + 1);
                    best_cost = cost;
                }
            }
        }
    }

    size_t build() {                                                                                              /* bvh.rs:257-324 */
        size_t node_count = 1;
        nodes[0].set_index(0);
        nodes[0].set_triangle_count((uint32_t)n_indices);
        update_node_aabb(0);
        std::vector<size_t> stack{0};
        while (!stack.empty()) {
            const size_t node_idx = stack.back();
            stack.pop_back();
            const Node node = nodes[node_idx];
            size_t best_axis;
            float best_split, best_cost;
            find_best_split_segmented(node, best_axis, best_split, best_cost);
            const float parent_cost = node.area() * (float)node.triangle_count();
            if (parent_cost <= best_cost) continue;
            /* (u32 arithmetic as written; `b` cannot pass below `a - 1`: the triangle holding bounds_min goes left) */
            uint32_t a = node.first_triangle_index();
            uint32_t b = a + node.triangle_count() - 1;
            while (a <= b) {
                const float centroid = centroids[a][best_axis];
                if (centroid < best_split) {
                    a += 1;
                } else {
                    const rpt_triangle ti = indices[a]; indices[a] = indices[b]; indices[b] = ti;
                    const Vec3 tc = centroids[a]; centroids[a] = centroids[b]; centroids[b] = tc;
                    if (b == 0) break;                             /* (Rust would panic on the underflow; unreachable, see above) */
                    b -= 1;
                }
            }
            const uint32_t left_count = a - node.first_triangle_index();
            if (left_count == 0 || left_count == node.triangle_count()) continue;
            const uint32_t prev_triangle_idx = node.first_triangle_index(), prev_triangle_count = node.triangle_count();
            const size_t left_idx = node_count, right_idx = node_count + 1;
            node_count += 2;
            nodes[node_idx].set_index((uint32_t)left_idx);
            nodes[node_idx].set_triangle_count(0);
            nodes[left_idx].set_index(prev_triangle_idx);
            nodes[left_idx].set_triangle_count(left_count);
            nodes[right_idx].set_index(a);
            nodes[right_idx].set_triangle_count(prev_triangle_count - left_count);
            update_node_aabb(left_idx);
            update_node_aabb(right_idx);
            stack.push_back(right_idx);
            stack.push_back(left_idx);
        }
        return node_count;
    }
};

}  // namespace

extern "C" {

/* BVHBuilder::new(vertices, indices).sah_samples(n).build(): reorders `triangles` in place, writes the node pool.
 * Returns 0, or -1 on bad arguments (nodes_capacity < 2 * n_triangles - 1, no triangles, an index out of range). */
int oracle_bvh_build(const float *vertices_xyzw, size_t n_vertices, rpt_triangle *triangles, size_t n_triangles, uint32_t sah_samples,
                     rpt_bvh_node *nodes_out, size_t nodes_capacity, size_t *n_nodes_out) {
    if (!vertices_xyzw || !triangles || !nodes_out || !n_nodes_out || n_triangles == 0 || sah_samples < 2 || nodes_capacity < 2 * n_triangles - 1) return -1;
    for (size_t i = 0; i < n_triangles; ++i)
        if (triangles[i].v0 >= n_vertices || triangles[i].v1 >= n_vertices || triangles[i].v2 >= n_vertices) return -1;
    Builder b;
    b.sah_samples = sah_samples;
    b.vertices = vertices_xyzw;
    b.indices = triangles;
    b.n_indices = n_triangles;
    b.centroids.resize(n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) {                                                                   /* bvh.rs:60-68 */
        const Vec3 s = v_add(v_add(b.vertex(triangles[i].v0), b.vertex(triangles[i].v1)), b.vertex(triangles[i].v2));
        b.centroids[i] = Vec3{s.x / 3.0f, s.y / 3.0f, s.z / 3.0f};
    }
    b.nodes.assign(2 * n_triangles - 1, Node());
    const size_t n = b.build();
    memcpy(nodes_out, b.nodes.data(), n * sizeof(Node));
    *n_nodes_out = n;
    return 0;
}

/* build_light_pick_table (reference src/light_pick.rs:24-122) with mask = compute_emissive_mask (:13-21), restated on its own (the product's
 * host mirror is csrc/host/light_table.cpp, its GPU form csrc/rpt_lights.hip; this is what both are checked against).  Writes at most `capacity`
 * entries; returns the number the table has (1 for the sentinel), or -1 on bad arguments.  Where the reference would panic — every emissive triangle
 * degenerate (bins[usize::MAX]), or the most probable cursor stepping below bin 0 — the loop stops instead. */
long oracle_light_table(const float *vertices_xyzw, size_t n_vertices, const rpt_triangle *tri, size_t n_triangles, const rpt_material_data *materials,
                        size_t n_materials, rpt_light_pick_entry *out, size_t capacity) {
    if (!vertices_xyzw || !tri || !materials || !out || capacity == 0) return -1;
    for (size_t i = 0; i < n_triangles; ++i)
        if (tri[i].v0 >= n_vertices || tri[i].v1 >= n_vertices || tri[i].v2 >= n_vertices || tri[i].material >= n_materials) return -1;
    auto P = [&](uint32_t v) { return Vec3{vertices_xyzw[4 * (size_t)v], vertices_xyzw[4 * (size_t)v + 1], vertices_xyzw[4 * (size_t)v + 2]}; };
    auto sub = [](Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto length = [](Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); };                            /* glam: dot(self).sqrt(), dot = x x + y y + z z left to right */
    std::vector<float> areas(n_triangles, 0.0f), powers(n_triangles, 0.0f), probabilities(n_triangles, 0.0f);
    float total_power = 0.0f;
    uint32_t total_tris = 0;
    for (size_t i = 0; i < n_triangles; ++i) {
        const float *em = materials[tri[i].material].emissive;
        if (!(em[0] != 0.0f || em[1] != 0.0f || em[2] != 0.0f)) continue;                                           /* :13-21, :36-38 */
        total_tris += 1;
        const Vec3 a = P(tri[i].v0), b = P(tri[i].v1), c = P(tri[i].v2);
        const Vec3 side_a = sub(b, a), side_b = sub(c, b), side_c = sub(a, c);                                      /* triangle_area, :5-11 */
        const float s = (length(side_a) + length(side_b) + length(side_c)) / 2.0f;
        const float area = std::sqrt(s * (s - length(side_a)) * (s - length(side_b)) * (s - length(side_c)));
        areas[i] = area;
        const float power = (em[0] * 1.0f + em[1] * 1.0f + em[2] * 1.0f) * area;                                    /* emissive.xyz().dot(Vec3::ONE) * area */
        powers[i] = power;
        total_power += power;
    }
    rpt_light_pick_entry sentinel{};
    sentinel.ratio = -1.0f;
    if (total_tris == 0) { out[0] = sentinel; return 1; }                                                           /* :52-58 */
    for (size_t i = 0; i < n_triangles; ++i) probabilities[i] = powers[i] / total_power;
    float sum = 0.0f;                                                                                               /* iter().sum::<f32>(): a fold from 0.0 */
    for (size_t i = 0; i < n_triangles; ++i) sum += probabilities[i];
    const float average_probability = sum / (float)total_tris;
    struct Bin { size_t index_a; float probability_a; size_t index_b; float probability_b; };
    std::vector<Bin> bins;
    for (size_t i = 0; i < n_triangles; ++i)
        if (probabilities[i] != 0.0f) bins.push_back(Bin{i, probabilities[i], 0, 0.0f});
    /* slice::sort_by is a stable merge sort; partial_cmp(..).unwrap_or(Equal) */
    std::stable_sort(bins.begin(), bins.end(), [](const Bin &x, const Bin &y) { return x.probability_a < y.probability_a; });
    const size_t num_bins = bins.size();
    if (num_bins == 0) { out[0] = sentinel; return 1; }
    size_t most_probable = num_bins - 1;
    for (size_t i = 0; i < num_bins; ++i) {                                                                         /* :89-104 */
        const float needed = average_probability - bins[i].probability_a;
        if (needed <= 0.0f) break;
        bins[i].index_b = bins[most_probable].index_a;
        bins[i].probability_b = needed;
        bins[most_probable].probability_a -= needed;
        if (bins[most_probable].probability_a <= average_probability) {
            if (most_probable == 0) break;
            most_probable -= 1;
        }
    }
    for (size_t i = 0; i < num_bins && i < capacity; ++i) {                                                         /* :106-119 */
        const Bin &x = bins[i];
        rpt_light_pick_entry e;
        e.triangle_index_a = (uint32_t)x.index_a;
        e.triangle_index_b = (uint32_t)x.index_b;
        e.triangle_pick_pdf_a = probabilities[x.index_a];
        e.triangle_area_a = areas[x.index_a];
        e.triangle_area_b = areas[x.index_b];
        e.triangle_pick_pdf_b = probabilities[x.index_b];
        e.ratio = x.probability_a / (x.probability_a + x.probability_b);
        out[i] = e;
    }
    return (long)num_bins;
}

}  // extern "C"
