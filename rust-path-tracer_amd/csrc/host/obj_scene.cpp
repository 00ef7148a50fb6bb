/*
 * obj_scene.cpp — Wavefront OBJ (+ MTL) into the World the kernels consume.
 *
 * The reference imports every format through assimp (src/asset.rs:55-72: JoinIdenticalVertices, Triangulate,
 * GenerateSmoothNormals, ...) and then treats all of them alike (asset.rs:78-128): positions and normals are
 * re-ordered (x, z, y), the winding becomes (f0, f2, f1), uv set 0 is taken as it comes, materials give
 * "$clr.diffuse" -> albedo, "$clr.emissive" x 15 -> emissive, "$mat.metallicFactor" / "$mat.roughnessFactor"
 * (asset.rs:157-170; assimp's MTL reader fills them from Kd, Ke, Pm, Pr).  This reader produces the same buffers from
 * the file directly: v / vt / vn / f (polygons fanned, negative indices), usemtl, mtllib with newmtl / Kd / Ke / Pm / Pr.
 * One output vertex per distinct (v, vt, vn) triple, in order of first use — assimp's join; its cache reordering and the
 * exact vertex order are not reproducible without it, so as for GLB parity is defined at the buffer boundary.
 * Without vn, normals are the angle-free smooth normals assimp generates: face normals summed over the faces that share
 * a POSITION, normalised.  Faces without a usemtl get the default material assimp appends (grey 0.6 diffuse).
 * map_Kd & co are not read (report, don't guess: the loader says so when a material names a texture).
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>

#include "host_internal.h"

namespace rpth {

namespace {

struct MtlEntry { float kd[3] = {0.6f, 0.6f, 0.6f}, ke[3] = {0, 0, 0}, pm = 0.0f, pr = 0.0f; bool has_ke = false, has_map = false; };

std::string dir_of(const std::string &path) {
    size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string(".") : path.substr(0, s);
}

bool read_text(const std::string &path, std::string &out) {
    std::vector<uint8_t> d;
    if (!read_file(path.c_str(), d)) return false;
    out.assign(d.begin(), d.end());
    return true;
}

void parse_mtl(const std::string &text, std::vector<std::string> &names, std::vector<MtlEntry> &mats) {
    size_t at = 0;
    MtlEntry *cur = nullptr;
    while (at < text.size()) {
        size_t e = text.find('\n', at);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(at, e - at);
        at = e + 1;
        char key[64] = {0}, name[512] = {0};
        float a, b, c;
        if (sscanf(line.c_str(), " newmtl %511s", name) == 1) {
            names.push_back(name);
            mats.emplace_back();
            cur = &mats.back();
        } else if (!cur) {
            continue;
        } else if (sscanf(line.c_str(), " Kd %f %f %f", &a, &b, &c) == 3) {
            cur->kd[0] = a; cur->kd[1] = b; cur->kd[2] = c;
        } else if (sscanf(line.c_str(), " Ke %f %f %f", &a, &b, &c) == 3) {
            cur->ke[0] = a; cur->ke[1] = b; cur->ke[2] = c; cur->has_ke = true;
        } else if (sscanf(line.c_str(), " Pm %f", &a) == 1) {
            cur->pm = a;
        } else if (sscanf(line.c_str(), " Pr %f", &a) == 1) {
            cur->pr = a;
        } else if (sscanf(line.c_str(), " %63s", key) == 1 && !strncmp(key, "map_", 4)) {
            cur->has_map = true;
        }
    }
}

}  // namespace

bool load_obj(const char *path, World &out) {
    std::string text;
    if (!read_text(path, text)) return false;
    std::vector<float> pos, nor, tex;                 /* file arrays, 3 / 3 / 2 per element */
    std::vector<std::string> mtl_names;
    std::vector<MtlEntry> mtl;
    struct Corner { long v, vt, vn; };
    struct Face { Corner c[3]; uint32_t material; };
    std::vector<Face> faces;
    bool used_default = false;
    long cur_mat = -1;
    size_t at = 0, line_no = 0;
    while (at < text.size()) {
        size_t e = text.find('\n', at);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(at, e - at);
        at = e + 1;
        ++line_no;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char *p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        float a, b, c;
        char name[512];
        auto blank = [](char ch) { return ch == ' ' || ch == '\t'; };   /* (a line that is exactly "vn" ends at p[2]: nothing beyond it is read) */
        if (p[0] == 'v' && blank(p[1]) && sscanf(p + 2, "%f %f %f", &a, &b, &c) == 3) {
            pos.insert(pos.end(), {a, b, c});
        } else if (p[0] == 'v' && p[1] == 'n' && blank(p[2]) && sscanf(p + 3, "%f %f %f", &a, &b, &c) == 3) {
            nor.insert(nor.end(), {a, b, c});
        } else if (p[0] == 'v' && p[1] == 't' && blank(p[2]) && sscanf(p + 3, "%f %f", &a, &b) >= 2) {
            tex.insert(tex.end(), {a, b});
        } else if (!strncmp(p, "mtllib ", 7) && sscanf(p + 7, "%511s", name) == 1) {
            std::string mt;
            if (read_text(dir_of(path) + "/" + name, mt)) parse_mtl(mt, mtl_names, mtl);   /* a missing library: default materials, as assimp */
        } else if (!strncmp(p, "usemtl ", 7) && sscanf(p + 7, "%511s", name) == 1) {
            cur_mat = -1;
            for (size_t i = 0; i < mtl_names.size(); ++i)
                if (mtl_names[i] == name) cur_mat = (long)i;
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<Corner> poly;
            const char *q = p + 1;
            while (*q) {
                while (*q == ' ' || *q == '\t') ++q;
                if (!*q) break;
                Corner cn{0, 0, 0};
                char *end = nullptr;
                cn.v = strtol(q, &end, 10);
                if (end == q) { set_error("OBJ: bad face on line " + std::to_string(line_no)); return false; }
                q = end;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') { cn.vt = strtol(q, &end, 10); q = end; }
                    if (*q == '/') { ++q; cn.vn = strtol(q, &end, 10); q = end; }
                }
                auto resolve = [](long i, size_t n) -> long { return i > 0 ? i - 1 : (i < 0 ? (long)n + i : -1); };
                const bool has_vt = cn.vt != 0, has_vn = cn.vn != 0;      /* 0 / absent: no such attribute on this corner */
                cn.v = resolve(cn.v, pos.size() / 3);
                cn.vt = resolve(cn.vt, tex.size() / 2);
                cn.vn = resolve(cn.vn, nor.size() / 3);
                if (cn.v < 0 || (size_t)cn.v >= pos.size() / 3 || cn.vt >= (long)(tex.size() / 2) || cn.vn >= (long)(nor.size() / 3) ||
                    (has_vt && cn.vt < 0) || (has_vn && cn.vn < 0)) {     /* (a relative index reaching before the first element) */
                    set_error("OBJ: index out of range on line " + std::to_string(line_no));
                    return false;
                }
                poly.push_back(cn);
            }
            if (poly.size() < 3) continue;
            uint32_t m;
            if (cur_mat >= 0) m = (uint32_t)cur_mat;
            else { used_default = true; m = 0xffffffffu; }
            for (size_t k = 1; k + 1 < poly.size(); ++k) faces.push_back(Face{{poly[0], poly[k], poly[k + 1]}, m});   /* Triangulate: fan */
        }
    }
    if (faces.empty()) { set_error("OBJ contains no faces"); return false; }
    for (const MtlEntry &m : mtl)
        if (m.has_map) { set_error("OBJ material references a texture map (map_*): texture files are only read from GLB"); return false; }

    /* smooth normals for corners without vn: sum of face normals over the faces sharing the position */
    std::vector<float> smooth;
    bool need_smooth = false;
    for (const Face &f : faces)
        for (const Corner &c : f.c) need_smooth |= c.vn < 0;
    if (need_smooth) {
        smooth.assign(pos.size(), 0.0f);
        for (const Face &f : faces) {
            const float *a = &pos[3 * f.c[0].v], *b = &pos[3 * f.c[1].v], *c = &pos[3 * f.c[2].v];
            float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            for (const Corner &cn : f.c)
                for (int k = 0; k < 3; ++k) smooth[3 * cn.v + k] += n[k];
        }
    }

    /* join identical (v, vt, vn) triples; emit in the reference's space: (x, z, y), winding (0, 2, 1) */
    std::vector<Vec4f> vertices, normals, tangents;
    std::vector<float> uvs;
    std::map<std::tuple<long, long, long>, uint32_t> seen;
    const uint32_t default_index = (uint32_t)mtl.size();
    auto vertex_of = [&](const Corner &c) -> uint32_t {
        auto key = std::make_tuple(c.v, c.vt, c.vn);
        auto it = seen.find(key);
        if (it != seen.end()) return it->second;
        uint32_t id = (uint32_t)vertices.size();
        seen.emplace(key, id);
        vertices.push_back(Vec4f{pos[3 * c.v], pos[3 * c.v + 2], pos[3 * c.v + 1], 1.0f});
        const float *n = c.vn >= 0 ? &nor[3 * c.vn] : &smooth[3 * c.v];
        float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        float inv = len > 0.0f ? 1.0f / len : 0.0f;
        normals.push_back(Vec4f{n[0] * inv, n[2] * inv, n[1] * inv, 0.0f});
        uvs.push_back(c.vt >= 0 ? tex[2 * c.vt] : 0.0f);
        uvs.push_back(c.vt >= 0 ? tex[2 * c.vt + 1] : 0.0f);
        return id;
    };
    out.indices.clear();
    for (const Face &f : faces) {
        uint32_t i0 = vertex_of(f.c[0]), i1 = vertex_of(f.c[1]), i2 = vertex_of(f.c[2]);
        out.indices.push_back(rpt_triangle{i0, i2, i1, f.material == 0xffffffffu ? default_index : f.material});
    }
    out.materials.assign(mtl.size() + (used_default ? 1 : 0), rpt_material_data{});
    for (size_t i = 0; i < out.materials.size(); ++i) {
        MtlEntry m = i < mtl.size() ? mtl[i] : MtlEntry{};
        rpt_material_data &d = out.materials[i];
        d.albedo[0] = m.kd[0]; d.albedo[1] = m.kd[1]; d.albedo[2] = m.kd[2]; d.albedo[3] = 1.0f;
        d.emissive[0] = m.ke[0] * 15.0f; d.emissive[1] = m.ke[1] * 15.0f; d.emissive[2] = m.ke[2] * 15.0f; d.emissive[3] = 1.0f * 15.0f;   /* asset.rs:163-166 */
        for (int k = 0; k < 4; ++k) { d.metallic[k] = m.pm; d.roughness[k] = m.pr; }
    }
    return finish_world(out, vertices, normals, tangents, uvs);
}

}  // namespace rpth
