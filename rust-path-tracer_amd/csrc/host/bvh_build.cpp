/*
 * bvh_build.cpp — binned-SAH BVH builder producing the reference's BVHNode pool.
 *
 * Behavioural restatement of src/bvh.rs (reference):
 *   BVHNodeExtensions (encapsulate / encapsulate_node / area)   bvh.rs:9-33
 *   BVHBuilder::new   (centroids = (v0+v1+v2)/3, 2N-1 node pool) bvh.rs:59-78
 *   update_node_aabb                                             bvh.rs:85-103
 *   find_best_split_segmented (sah_samples equal-width bins)     bvh.rs:178-255
 *   build (explicit stack, in-place two-pointer partition,
 *          children allocated consecutively, left first)         bvh.rs:257-324
 * The node ORDER and the in-place triangle reorder are part of the buffer
 * contract (leaf ranges, light table and MIS compare index the reordered
 * index buffer), so the float arithmetic follows the reference's f32 steps.
 */
#include <cmath>
#include <cstring>
#include <limits>

#include "host_internal.h"

namespace rpth {
namespace {

struct V3 { float x, y, z; };
inline float fmin_r(float a, float b) { return (a < b || b != b) ? a : b; }   /* f32::min */
inline float fmax_r(float a, float b) { return (a > b || b != b) ? a : b; }   /* f32::max */
inline V3 vmin(V3 a, V3 b) { return V3{fmin_r(a.x, b.x), fmin_r(a.y, b.y), fmin_r(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{fmax_r(a.x, b.x), fmax_r(a.y, b.y), fmax_r(a.z, b.z)}; }
inline V3 xyz(const Vec4f &v) { return V3{v.x, v.y, v.z}; }
inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

const float INF = std::numeric_limits<float>::infinity();

struct Box {   /* BVHNode used as a plain AABB (bvh.rs:9-33), default = inverted infinite box */
    V3 mn{INF, INF, INF}, mx{-INF, -INF, -INF};
    void encapsulate(V3 p) { mn = vmin(mn, p); mx = vmax(mx, p); }
    void encapsulate_box(const Box &b) {
        if (b.mn.x == INF) return;
        mn = vmin(mn, b.mn);
        mx = vmax(mx, b.mx);
    }
    float area() const {
        V3 e{mx.x - mn.x, mx.y - mn.y, mx.z - mn.z};
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
};

inline float node_area(const rpt_bvh_node &n) {
    V3 e{n.aabb_max[0] - n.aabb_min[0], n.aabb_max[1] - n.aabb_min[1], n.aabb_max[2] - n.aabb_min[2]};
    return e.x * e.y + e.y * e.z + e.z * e.x;
}

/* Rust `f32 as usize`: saturating, NaN -> 0 */
inline size_t f2usize(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return ~(size_t)0;
    return (size_t)x;
}

struct Builder {
    const Vec4f *vertices;
    rpt_triangle *indices;
    size_t n_triangles;
    uint32_t sah_samples;
    std::vector<V3> centroids;
    std::vector<rpt_bvh_node> &nodes;

    void update_node_aabb(size_t node_idx) {
        rpt_bvh_node &node = nodes[node_idx];
        V3 mn{INF, INF, INF}, mx{-INF, -INF, -INF};
        for (uint32_t i = 0; i < node.triangle_count; ++i) {
            const rpt_triangle &t = indices[node.left_or_first + i];
            V3 v0 = xyz(vertices[t.v0]), v1 = xyz(vertices[t.v1]), v2 = xyz(vertices[t.v2]);
            mn = vmin(mn, vmin(vmin(v0, v1), v2));
            mx = vmax(mx, vmax(vmax(v0, v1), v2));
        }
        node.aabb_min[0] = mn.x; node.aabb_min[1] = mn.y; node.aabb_min[2] = mn.z;
        node.aabb_max[0] = mx.x; node.aabb_max[1] = mx.y; node.aabb_max[2] = mx.z;
    }

    void find_best_split_segmented(const rpt_bvh_node &node, int &best_axis, float &best_split, float &best_cost) {
        best_axis = 0;
        best_split = 0.0f;
        best_cost = INF;
        const uint32_t S = sah_samples;
        std::vector<Box> seg_box(S);
        std::vector<uint32_t> seg_count(S);
        std::vector<float> left_areas(S - 1), right_areas(S - 1);
        std::vector<uint32_t> left_counts(S - 1), right_counts(S - 1);
        for (int axis = 0; axis < 3; ++axis) {
            float bounds_min = INF, bounds_max = -INF;
            for (uint32_t i = 0; i < node.triangle_count; ++i) {
                float c = axis_of(centroids[node.left_or_first + i], axis);
                bounds_min = fmin_r(bounds_min, c);
                bounds_max = fmax_r(bounds_max, c);
            }
            if (bounds_min == bounds_max) continue;

            for (uint32_t s = 0; s < S; ++s) { seg_box[s] = Box(); seg_count[s] = 0; }
            float scale = (float)S / (bounds_max - bounds_min);
            for (uint32_t i = 0; i < node.triangle_count; ++i) {
                size_t ti = node.left_or_first + i;
                const rpt_triangle &t = indices[ti];
                V3 v0 = xyz(vertices[t.v0]), v1 = xyz(vertices[t.v1]), v2 = xyz(vertices[t.v2]);
                size_t si = f2usize((axis_of(centroids[ti], axis) - bounds_min) * scale);
                if (si > S - 1) si = S - 1;
                seg_box[si].encapsulate(v0);
                seg_box[si].encapsulate(v1);
                seg_box[si].encapsulate(v2);
                seg_count[si] += 1;
            }

            Box left_box, right_box;
            uint32_t left_sum = 0, right_sum = 0;
            for (uint32_t i = 0; i < S - 1; ++i) {
                left_sum += seg_count[i];
                left_counts[i] = left_sum;
                left_box.encapsulate_box(seg_box[i]);
                left_areas[i] = left_box.area();
                right_sum += seg_count[S - 1 - i];
                right_counts[S - 2 - i] = right_sum;
                right_box.encapsulate_box(seg_box[S - 1 - i]);
                right_areas[S - 2 - i] = right_box.area();
            }

            float scale2 = (bounds_max - bounds_min) / (float)S;
            for (uint32_t i = 0; i < S - 1; ++i) {
                float cost = (float)left_counts[i] * left_areas[i] + (float)right_counts[i] * right_areas[i];
                if (cost < best_cost) {
                    best_axis = axis;
                    best_split = bounds_min + scale2 * (float)(i + 1);
                    best_cost = cost;
                }
            }
        }
    }

    size_t build() {
        size_t node_count = 1;
        nodes[0].left_or_first = 0;
        nodes[0].triangle_count = (uint32_t)n_triangles;
        update_node_aabb(0);

        std::vector<size_t> stack{0};
        while (!stack.empty()) {
            size_t node_idx = stack.back();
            stack.pop_back();
            const rpt_bvh_node node = nodes[node_idx];

            int best_axis;
            float best_split, best_cost;
            find_best_split_segmented(node, best_axis, best_split, best_cost);

            float parent_cost = node_area(node) * (float)node.triangle_count;
            if (parent_cost <= best_cost) continue;

            /* two-pointer partition (bvh.rs:281-292); signed to survive b -> -1 */
            int64_t a = node.left_or_first;
            int64_t b = a + (int64_t)node.triangle_count - 1;
            while (a <= b) {
                float c = axis_of(centroids[(size_t)a], best_axis);
                if (c < best_split) {
                    a += 1;
                } else {
                    std::swap(indices[(size_t)a], indices[(size_t)b]);
                    std::swap(centroids[(size_t)a], centroids[(size_t)b]);
                    b -= 1;
                }
            }
            uint32_t left_count = (uint32_t)(a - (int64_t)node.left_or_first);
            if (left_count == 0 || left_count == node.triangle_count) continue;

            size_t left_idx = node_count, right_idx = node_count + 1;
            node_count += 2;
            nodes[node_idx].left_or_first = (uint32_t)left_idx;
            nodes[node_idx].triangle_count = 0;
            nodes[left_idx].left_or_first = node.left_or_first;
            nodes[left_idx].triangle_count = left_count;
            nodes[right_idx].left_or_first = (uint32_t)a;
            nodes[right_idx].triangle_count = node.triangle_count - left_count;
            update_node_aabb(left_idx);
            update_node_aabb(right_idx);
            stack.push_back(right_idx);
            stack.push_back(left_idx);
        }
        return node_count;
    }
};

}  // namespace

size_t bvh_build(const Vec4f *vertices, rpt_triangle *triangles, size_t n_triangles, uint32_t sah_samples,
                 std::vector<rpt_bvh_node> &nodes) {
    rpt_bvh_node def;
    def.aabb_min[0] = def.aabb_min[1] = def.aabb_min[2] = INF;
    def.aabb_max[0] = def.aabb_max[1] = def.aabb_max[2] = -INF;
    def.triangle_count = 0;
    def.left_or_first = 0;
    nodes.assign(n_triangles * 2 - 1, def);
    if (sah_samples < 2) sah_samples = 2;
    Builder b{vertices, triangles, n_triangles, sah_samples, {}, nodes};
    b.centroids.resize(n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) {
        V3 v0 = xyz(vertices[triangles[i].v0]), v1 = xyz(vertices[triangles[i].v1]), v2 = xyz(vertices[triangles[i].v2]);
        /* (v0 + v1 + v2) / 3.0 */
        b.centroids[i] = V3{((v0.x + v1.x) + v2.x) / 3.0f, ((v0.y + v1.y) + v2.y) / 3.0f, ((v0.z + v1.z) + v2.z) / 3.0f};
    }
    size_t n = b.build();
    nodes.resize(n);
    return n;
}

/* Depth of the tree under node 0; 0xffffffff if the array is NOT a tree (a node reached more often than there are nodes: a
 * cycle or shared children in a file that did not come from the builder — an .rptscene cache is untrusted input, and an
 * unbounded walk here was a hang found by tools/fuzz_glb.py) or a child index leaves the array. */
uint32_t bvh_max_depth(const std::vector<rpt_bvh_node> &nodes) {
    if (nodes.empty()) return 0;
    uint32_t best = 0;
    size_t visited = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 0u}};
    while (!st.empty()) {
        auto [idx, d] = st.back();
        st.pop_back();
        if (++visited > nodes.size() || idx >= nodes.size()) return 0xffffffffu;
        if (d > best) best = d;
        const rpt_bvh_node &n = nodes[idx];
        if (n.triangle_count == 0) {
            st.push_back({n.left_or_first, d + 1});
            st.push_back({n.left_or_first + 1, d + 1});
        }
    }
    return best;
}

}  // namespace rpth
