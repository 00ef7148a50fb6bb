/*
 * image_io.cpp — the two data formats either side of the hot path:
 *   - PNG output of a resolved frame, as the reference's "Save render" produces it (src/app.rs:759-845: the
 *     tonemapped colour goes through the sRGB-encoding surface format, 8 bits per channel, alpha 255);
 *   - ".rptscene": the five POD buffers of a World (src/asset.rs:9-16) + optional atlas in one flat file, so
 *     the Rust host and this backend can exchange byte-identical inputs (closes the "no assimp here" gap).
 */
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../rpt_math.h"
#include "host_internal.h"

using namespace rpth;

namespace {
void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void png_chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &body) {
    put_be32(out, (uint32_t)body.size());
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put_be32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(body.size() + 4)));
}
/* linear -> sRGB transfer function (IEC 61966-2-1), as an sRGB render target applies it */
float srgb_encode(float x) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    return x <= 0.0031308f ? 12.92f * x : 1.055f * rptm::powr(x, 1.0f / 2.4f) - 0.055f;
}

const char SCENE_MAGIC[8] = {'R', 'P', 'T', 'S', 'C', 'N', '0', '1'};
struct SceneHeader {
    char magic[8];
    uint64_t n_vertices, n_triangles, n_nodes, n_materials, n_light_pick;
    uint32_t atlas_w, atlas_h;
};
}  // namespace

extern "C" {

int rpt_write_png(const char *path, const float *rgb, uint32_t width, uint32_t height, int srgb_encode_flag) {
    if (!path || !rgb || !width || !height) { set_error("null/empty argument"); return -1; }
    std::vector<uint8_t> raw((size_t)height * (1 + (size_t)width * 4));
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t *row = &raw[(size_t)y * (1 + (size_t)width * 4)];
        row[0] = 0;   /* filter: none */
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 4; ++c) {
                float v = c < 3 ? rgb[((size_t)y * width + x) * 3 + c] : 1.0f;
                if (c < 3 && srgb_encode_flag) v = srgb_encode(v);
                v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
                row[1 + x * 4 + c] = (uint8_t)(v * 255.0f + 0.5f);
            }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_error("deflate failed"); return -1; }
    z.resize(zlen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   /* RGBA8 */
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    FILE *f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return -1; }
    size_t w = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    if (w != out.size()) { set_error("short write"); return -1; }
    return 0;
}

int rpt_world_save(const rpt_world *world, const char *path) {
    if (!world || !path) { set_error("null argument"); return -1; }
    const World &w = world->w;
    SceneHeader h{};
    memcpy(h.magic, SCENE_MAGIC, 8);
    h.n_vertices = w.per_vertex.size(); h.n_triangles = w.indices.size(); h.n_nodes = w.nodes.size();
    h.n_materials = w.materials.size(); h.n_light_pick = w.light_pick.size();
    h.atlas_w = w.atlas.empty() ? 0 : w.atlas_w; h.atlas_h = w.atlas.empty() ? 0 : w.atlas_h;
    FILE *f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return -1; }
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    auto put = [&](const void *p, size_t bytes) { if (ok && bytes) ok = fwrite(p, 1, bytes, f) == bytes; };
    put(w.per_vertex.data(), w.per_vertex.size() * sizeof(rpt_per_vertex_data));
    put(w.indices.data(), w.indices.size() * sizeof(rpt_triangle));
    put(w.nodes.data(), w.nodes.size() * sizeof(rpt_bvh_node));
    put(w.materials.data(), w.materials.size() * sizeof(rpt_material_data));
    put(w.light_pick.data(), w.light_pick.size() * sizeof(rpt_light_pick_entry));
    put(w.atlas.data(), w.atlas.size());
    fclose(f);
    if (!ok) { set_error("short write"); return -1; }
    return 0;
}

int rpt_world_load_cache(const char *path, rpt_world **out) {
    if (!path || !out) { set_error("null argument"); return -1; }
    FILE *f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return RPT_HOST_ELOAD; }
    SceneHeader h{};
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && !memcmp(h.magic, SCENE_MAGIC, 8);
    const uint64_t LIMIT = 1ull << 31;
    ok = ok && h.n_vertices && h.n_triangles && h.n_nodes && h.n_materials && h.n_light_pick && h.n_vertices < LIMIT &&
         h.n_triangles < LIMIT && h.n_nodes < LIMIT && h.n_materials < LIMIT && h.n_light_pick < LIMIT && h.atlas_w <= 65536 &&
         h.atlas_h <= 65536;
    if (!ok) { fclose(f); set_error("not an .rptscene file"); return RPT_HOST_ELOAD; }
    {
        /* the counts of an untrusted header size nothing before the FILE is known to hold that many records (a mutated count of 2^31 - 1 vertices
           used to ask for 128 GB: std::bad_alloc across the C ABI; found by tools/fuzz_glb.py under AddressSanitizer, round 4) */
        const uint64_t need = sizeof(h) + h.n_vertices * sizeof(rpt_per_vertex_data) + h.n_triangles * sizeof(rpt_triangle) + h.n_nodes * sizeof(rpt_bvh_node) +
                              h.n_materials * sizeof(rpt_material_data) + h.n_light_pick * sizeof(rpt_light_pick_entry) + (uint64_t)h.atlas_w * h.atlas_h * 4u;
        long here = ftell(f);
        bool sized = here >= 0 && fseek(f, 0, SEEK_END) == 0;
        const long end = sized ? ftell(f) : -1;
        sized = sized && end >= 0 && fseek(f, here, SEEK_SET) == 0;
        if (!sized || (uint64_t)end < need) { fclose(f); set_error("truncated or inconsistent .rptscene file"); return RPT_HOST_ELOAD; }
    }
    rpt_world *w = nullptr;
    try {
        w = new rpt_world();
        World &d0 = w->w;
        d0.per_vertex.resize(h.n_vertices); d0.indices.resize(h.n_triangles); d0.nodes.resize(h.n_nodes);
        d0.materials.resize(h.n_materials); d0.light_pick.resize(h.n_light_pick);
        d0.atlas.resize((size_t)h.atlas_w * h.atlas_h * 4);
    } catch (const std::exception &) {
        delete w;
        fclose(f);
        set_error("out of memory loading the .rptscene file");
        return RPT_HOST_ELOAD;
    }
    World &d = w->w;
    d.atlas_w = h.atlas_w; d.atlas_h = h.atlas_h;
    auto get = [&](void *p, size_t bytes) { if (ok && bytes) ok = fread(p, 1, bytes, f) == bytes; };
    get(d.per_vertex.data(), d.per_vertex.size() * sizeof(rpt_per_vertex_data));
    get(d.indices.data(), d.indices.size() * sizeof(rpt_triangle));
    get(d.nodes.data(), d.nodes.size() * sizeof(rpt_bvh_node));
    get(d.materials.data(), d.materials.size() * sizeof(rpt_material_data));
    get(d.light_pick.data(), d.light_pick.size() * sizeof(rpt_light_pick_entry));
    get(d.atlas.data(), d.atlas.size());
    fclose(f);
    for (const rpt_triangle &t : d.indices)
        if (ok && (t.v0 >= h.n_vertices || t.v1 >= h.n_vertices || t.v2 >= h.n_vertices || t.material >= h.n_materials)) ok = false;
    for (const rpt_bvh_node &n : d.nodes)
        if (ok && (n.triangle_count ? (uint64_t)n.left_or_first + n.triangle_count > h.n_triangles : (uint64_t)n.left_or_first + 1 >= h.n_nodes)) ok = false;
    for (const rpt_light_pick_entry &e : d.light_pick)
        if (ok && !(d.light_pick[0].ratio < 0.0f) && (e.triangle_index_a >= h.n_triangles || e.triangle_index_b >= h.n_triangles)) ok = false;
    if (ok) {
        d.max_depth = bvh_max_depth(d.nodes);
        if (d.max_depth == 0xffffffffu) ok = false;           /* a cycle, shared children: not a tree */
    }
    if (!ok) { delete w; set_error("truncated or inconsistent .rptscene file"); return RPT_HOST_ELOAD; }
    d.n_emissive = 0;
    for (const rpt_triangle &t : d.indices) {
        const float *e = d.materials[t.material].emissive;
        if (e[0] != 0.0f || e[1] != 0.0f || e[2] != 0.0f) d.n_emissive += 1;
    }
    *out = w;
    return 0;
}

}  // extern "C"
