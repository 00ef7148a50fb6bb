/*
 * light_table.cpp — power-weighted alias ("robin hood") light pick table.
 *
 * Behavioural restatement of src/light_pick.rs (reference):
 *   triangle_area (Heron)                 light_pick.rs:5-11
 *   compute_emissive_mask                 light_pick.rs:13-21
 *   build_light_pick_table                light_pick.rs:24-122
 * Output layout: LightPickEntry (shared_structs/src/lib.rs:102-119); a scene
 * without emissive triangles yields the one-entry sentinel (ratio = -1).
 */
#include <algorithm>
#include <cmath>

#include "host_internal.h"

namespace rpth {
namespace {
struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float len(V3 a) { return std::sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
inline V3 xyz(const Vec4f &v) { return V3{v.x, v.y, v.z}; }

float triangle_area(V3 a, V3 b, V3 c) {
    V3 side_a = sub(b, a), side_b = sub(c, b), side_c = sub(a, c);
    float s = (len(side_a) + len(side_b) + len(side_c)) / 2.0f;
    return std::sqrt(s * (s - len(side_a)) * (s - len(side_b)) * (s - len(side_c)));
}
}  // namespace

std::vector<rpt_light_pick_entry> build_light_pick_table(const Vec4f *vertices, const rpt_triangle *triangles,
                                                         size_t n_triangles, const rpt_material_data *materials,
                                                         uint32_t *n_emissive) {
    std::vector<float> areas(n_triangles, 0.0f), powers(n_triangles, 0.0f);
    float total_power = 0.0f;
    uint32_t total_tris = 0;
    for (size_t i = 0; i < n_triangles; ++i) {
        const float *e = materials[triangles[i].material].emissive;
        bool mask = (e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f);   /* compute_emissive_mask */
        if (!mask) continue;
        total_tris += 1;
        V3 a = xyz(vertices[triangles[i].v0]), b = xyz(vertices[triangles[i].v1]), c = xyz(vertices[triangles[i].v2]);
        float area = triangle_area(a, b, c);
        areas[i] = area;
        float power = ((e[0] * 1.0f) + (e[1] * 1.0f) + (e[2] * 1.0f)) * area;   /* emissive.dot(Vec3::ONE) * area */
        powers[i] = power;
        total_power += power;
    }
    if (n_emissive) *n_emissive = total_tris;
    if (total_tris == 0) {
        rpt_light_pick_entry s{};
        s.ratio = -1.0f;
        return {s};
    }
    std::vector<float> prob(n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) prob[i] = powers[i] / total_power;
    float sum = 0.0f;
    for (size_t i = 0; i < n_triangles; ++i) sum += prob[i];
    float average_probability = sum / (float)total_tris;

    struct Bin { size_t index_a; float probability_a; size_t index_b; float probability_b; };
    std::vector<Bin> bins;
    for (size_t i = 0; i < n_triangles; ++i)
        if (prob[i] != 0.0f) bins.push_back(Bin{i, prob[i], 0, 0.0f});
    /* Rust slice::sort_by is stable; partial_cmp -> Equal on NaN */
    std::stable_sort(bins.begin(), bins.end(), [](const Bin &a, const Bin &b) { return a.probability_a < b.probability_a; });

    size_t num_bins = bins.size();
    if (num_bins == 0) {   /* all emissive triangles degenerate: the reference would index bins[usize::MAX] and panic */
        rpt_light_pick_entry s{};
        s.ratio = -1.0f;
        return {s};
    }
    size_t most_probable = num_bins - 1;
    for (size_t i = 0; i < num_bins; ++i) {
        float needed = average_probability - bins[i].probability_a;
        if (needed <= 0.0f) break;
        bins[i].index_b = bins[most_probable].index_a;
        bins[i].probability_b = needed;
        bins[most_probable].probability_a -= needed;
        if (bins[most_probable].probability_a <= average_probability) {
            if (most_probable == 0) break;   /* reference: usize underflow panic */
            most_probable -= 1;
        }
    }

    std::vector<rpt_light_pick_entry> table;
    table.reserve(num_bins);
    for (const Bin &x : bins) {
        rpt_light_pick_entry e;
        e.triangle_index_a = (uint32_t)x.index_a;
        e.triangle_index_b = (uint32_t)x.index_b;
        e.triangle_pick_pdf_a = prob[x.index_a];
        e.triangle_area_a = areas[x.index_a];
        e.triangle_area_b = areas[x.index_b];
        e.triangle_pick_pdf_b = prob[x.index_b];
        e.ratio = x.probability_a / (x.probability_a + x.probability_b);
        table.push_back(e);
    }
    return table;
}

}  // namespace rpth
