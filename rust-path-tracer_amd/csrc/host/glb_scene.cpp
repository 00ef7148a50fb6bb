/*
 * glb_scene.cpp — scene import for the render dispatch: glTF-binary -> World.
 *
 * Restates World::from_path (reference src/asset.rs:55-224) for .glb input.
 * The reference imports through assimp 5.2.5 (russimp), which is not available
 * here, so the glTF side is a small self-contained reader (JSON chunk + BIN
 * chunk; node TRS/matrix; POSITION / NORMAL / TEXCOORD_0; u8/u16/u32 indices;
 * pbrMetallicRoughness + emissiveFactor) emitting what assimp's glTF2 importer
 * hands to asset.rs:
 *   - one mesh per primitive, nodes walked depth-first, meshes before children
 *     (asset.rs:97-127); several scene roots hang under a synthetic identity root;
 *   - world transform applied, then the (x, z, y) axis swap (asset.rs:101-102)
 *     and the (i0, i2, i1) winding swap with material index in .w (asset.rs:106);
 *   - normals: rotate (n / scale) by the node's rotation, normalise, swap
 *     (asset.rs:108-111);
 *   - TEXCOORD_0 with assimp's v -> 1 - v flip; missing UVs zero-filled
 *     (asset.rs:116-122);
 *   - materials: baseColorFactor -> albedo, emissiveFactor (alpha 1) * 15 ->
 *     emissive, metallicFactor / roughnessFactor (glTF default 1.0) splatted
 *     (asset.rs:162-174); a primitive without a material gets an appended
 *     default material, as assimp does.
 * Not reproducible without assimp (stated in DESIGN.md): JoinIdenticalVertices
 * / ImproveCacheLocality reorderings (asset.rs:59,67) and CalculateTangentSpace
 * tangents (left zero; only read when a normal texture exists, and no shipped
 * scene has textures).  Texture atlas packing (src/atlas.rs) is out of scope.
 */
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>

#include "host_internal.h"

namespace rpth {
namespace {

/* ---------------------------------------------------------------- JSON -- */
struct JValue;
using JPtr = std::shared_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;
    const JValue *get(const char *key) const {
        for (auto &kv : obj) if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    double number_or(const char *key, double dflt) const {
        const JValue *v = get(key);
        return (v && v->kind == Num) ? v->num : dflt;
    }
};

struct JParser {
    const char *p, *end;
    bool ok = true;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    JPtr fail() { ok = false; return std::make_shared<JValue>(); }
    JPtr parse() {
        ws();
        if (p >= end) return fail();
        auto v = std::make_shared<JValue>();
        char c = *p;
        if (c == '{') {
            v->kind = JValue::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return v; }
            while (ok) {
                ws();
                JPtr k = parse();
                if (!ok || k->kind != JValue::Str) return fail();
                ws();
                if (p >= end || *p != ':') return fail();
                ++p;
                JPtr val = parse();
                v->obj.emplace_back(k->str, val);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                return fail();
            }
        } else if (c == '[') {
            v->kind = JValue::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return v; }
            while (ok) {
                v->arr.push_back(parse());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                return fail();
            }
        } else if (c == '"') {
            v->kind = JValue::Str;
            ++p;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) {
                    ++p;
                    switch (*p) {
                        case 'n': v->str.push_back('\n'); break;
                        case 't': v->str.push_back('\t'); break;
                        case 'u': v->str.push_back('?'); p += 4; break;
                        default: v->str.push_back(*p);
                    }
                    ++p;
                } else {
                    v->str.push_back(*p++);
                }
            }
            if (p >= end) return fail();
            ++p;
        } else if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) {
            v->kind = JValue::Bool; v->b = true; p += 4;
        } else if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) {
            v->kind = JValue::Bool; v->b = false; p += 5;
        } else if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) {
            p += 4;
        } else {
            char *q = nullptr;
            std::string tmp(p, (size_t)std::min<ptrdiff_t>(end - p, 64));
            v->kind = JValue::Num;
            v->num = strtod(tmp.c_str(), &q);
            if (q == tmp.c_str()) return fail();
            p += (q - tmp.c_str());
        }
        return v;
    }
};

/* ------------------------------------------------- small float matrices -- */
struct M4 { float m[4][4]; };   /* m[col][row], glam column-major */
M4 identity() {
    M4 r{};
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
    return r;
}
M4 mul(const M4 &a, const M4 &b) {
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.m[k][row] * b.m[c][k];
            r.m[c][row] = s;
        }
    return r;
}
struct Quat { float x, y, z, w; };
M4 from_trs(const float t[3], Quat q, const float s[3]) {
    /* rotation matrix of a unit quaternion, columns scaled, translation appended */
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    M4 r = identity();
    r.m[0][0] = (1.0f - (yy + zz)) * s[0]; r.m[0][1] = (xy + wz) * s[0]; r.m[0][2] = (xz - wy) * s[0];
    r.m[1][0] = (xy - wz) * s[1]; r.m[1][1] = (1.0f - (xx + zz)) * s[1]; r.m[1][2] = (yz + wx) * s[1];
    r.m[2][0] = (xz + wy) * s[2]; r.m[2][1] = (yz - wx) * s[2]; r.m[2][2] = (1.0f - (xx + yy)) * s[2];
    r.m[3][0] = t[0]; r.m[3][1] = t[1]; r.m[3][2] = t[2];
    return r;
}
/* Mat4::to_scale_rotation_translation: scale = column lengths (x negated when
 * det < 0), rotation = quaternion of the normalised axes */
void decompose(const M4 &m, float scale[3], Quat &q) {
    auto col_len = [&](int c) { return std::sqrt(m.m[c][0] * m.m[c][0] + m.m[c][1] * m.m[c][1] + m.m[c][2] * m.m[c][2]); };
    float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[2][1] * m.m[1][2]) -
                m.m[1][0] * (m.m[0][1] * m.m[2][2] - m.m[2][1] * m.m[0][2]) +
                m.m[2][0] * (m.m[0][1] * m.m[1][2] - m.m[1][1] * m.m[0][2]);
    scale[0] = col_len(0) * (det < 0.0f ? -1.0f : 1.0f);
    scale[1] = col_len(1);
    scale[2] = col_len(2);
    float r[3][3];
    for (int c = 0; c < 3; ++c)
        for (int row = 0; row < 3; ++row) r[c][row] = m.m[c][row] / scale[c];
    /* Quat::from_rotation_axes */
    float m00 = r[0][0], m01 = r[0][1], m02 = r[0][2];
    float m10 = r[1][0], m11 = r[1][1], m12 = r[1][2];
    float m20 = r[2][0], m21 = r[2][1], m22 = r[2][2];
    if (m22 <= 0.0f) {
        float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) {
            float four_xsq = omm22 - dif10, inv4x = 0.5f / std::sqrt(four_xsq);
            q = Quat{four_xsq * inv4x, (m01 + m10) * inv4x, (m02 + m20) * inv4x, (m12 - m21) * inv4x};
        } else {
            float four_ysq = omm22 + dif10, inv4y = 0.5f / std::sqrt(four_ysq);
            q = Quat{(m01 + m10) * inv4y, four_ysq * inv4y, (m12 + m21) * inv4y, (m20 - m02) * inv4y};
        }
    } else {
        float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) {
            float four_zsq = opm22 - sum10, inv4z = 0.5f / std::sqrt(four_zsq);
            q = Quat{(m02 + m20) * inv4z, (m12 + m21) * inv4z, four_zsq * inv4z, (m01 - m10) * inv4z};
        } else {
            float four_wsq = opm22 + sum10, inv4w = 0.5f / std::sqrt(four_wsq);
            q = Quat{(m12 - m21) * inv4w, (m20 - m02) * inv4w, (m01 - m10) * inv4w, four_wsq * inv4w};
        }
    }
}
/* Quat::mul_vec3: v*(w^2 - b.b) + b*(2 v.b) + (b x v)*(2w) */
void quat_rotate(Quat q, const float v[3], float out[3]) {
    float b[3] = {q.x, q.y, q.z};
    float b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    float vb = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
    float cx = b[1] * v[2] - b[2] * v[1], cy = b[2] * v[0] - b[0] * v[2], cz = b[0] * v[1] - b[1] * v[0];
    float k0 = q.w * q.w - b2, k1 = 2.0f * vb, k2 = 2.0f * q.w;
    out[0] = v[0] * k0 + b[0] * k1 + cx * k2;
    out[1] = v[1] * k0 + b[1] * k1 + cy * k2;
    out[2] = v[2] * k0 + b[2] * k1 + cz * k2;
}

/* ---------------------------------------------------------- accessors -- */
struct Glb {
    JPtr root;
    std::vector<uint8_t> bin;
    const JValue *arr(const char *name) const {
        const JValue *v = root->get(name);
        return (v && v->kind == JValue::Arr) ? v : nullptr;
    }
};

bool read_accessor(const Glb &g, int index, int want_components, std::vector<double> &out, size_t &count) {
    const JValue *accs = g.arr("accessors"), *views = g.arr("bufferViews");
    if (!accs || !views || index < 0 || (size_t)index >= accs->arr.size()) return false;
    const JValue &a = *accs->arr[index];
    int view_idx = (int)a.number_or("bufferView", -1);
    if (view_idx < 0 || (size_t)view_idx >= views->arr.size()) return false;
    const JValue &bv = *views->arr[view_idx];
    /* every number comes from the file: convert and bound it before it sizes or addresses anything (a crafted count
     * must not reach resize(), offset + i * stride must not wrap) */
    auto to_size = [](double v, size_t &dst) {
        if (!(v >= 0.0) || v > 4.0e15 || v != std::floor(v)) return false;
        dst = (size_t)v;
        return true;
    };
    size_t view_off = 0, acc_off = 0, stride = 0;
    if (!to_size(bv.number_or("byteOffset", 0), view_off) || !to_size(a.number_or("byteOffset", 0), acc_off) ||
        !to_size(bv.number_or("byteStride", 0), stride) || !to_size(a.number_or("count", 0), count))
        return false;
    const size_t offset = view_off + acc_off;
    int ctype = (int)a.number_or("componentType", 0);
    const JValue *type = a.get("type");
    if (!type) return false;
    int comps = type->str == "SCALAR" ? 1 : type->str == "VEC2" ? 2 : type->str == "VEC3" ? 3 : type->str == "VEC4" ? 4 : 0;
    if (comps == 0 || (want_components && comps != want_components)) return false;
    size_t csize = (ctype == 5126 || ctype == 5125) ? 4 : (ctype == 5123 || ctype == 5122) ? 2 : (ctype == 5121 || ctype == 5120) ? 1 : 0;
    if (!csize) return false;
    if (!stride) stride = csize * comps;
    bool normalized = a.get("normalized") && a.get("normalized")->b;
    const size_t elem = csize * (size_t)comps, bin = g.bin.size();
    if (stride > (1u << 20) || offset > bin) return false;
    if (count != 0 && (elem > bin - offset || (count - 1) > (bin - offset - elem) / stride)) return false;   /* last element ends inside the chunk */
    out.resize(count * comps);
    for (size_t i = 0; i < count; ++i)
        for (int c = 0; c < comps; ++c) {
            size_t at = offset + i * stride + (size_t)c * csize;
            if (at + csize > g.bin.size()) return false;
            const uint8_t *p = g.bin.data() + at;
            double v;
            switch (ctype) {
                case 5126: { float f; memcpy(&f, p, 4); v = f; break; }
                case 5125: { uint32_t u; memcpy(&u, p, 4); v = u; break; }
                case 5123: { uint16_t u; memcpy(&u, p, 2); v = normalized ? u / 65535.0 : u; break; }
                case 5122: { int16_t u; memcpy(&u, p, 2); v = normalized ? std::max(u / 32767.0, -1.0) : u; break; }
                case 5121: { v = normalized ? *p / 255.0 : *p; break; }
                default: { int8_t s; memcpy(&s, p, 1); v = normalized ? std::max(s / 127.0, -1.0) : s; break; }
            }
            out[i * comps + c] = v;
        }
    return true;
}

struct Gather {
    std::vector<Vec4f> vertices, normals, tangents;
    std::vector<float> uvs;   /* 2 per vertex */
    bool any_tangent_missing = false;
    std::vector<rpt_triangle> indices;
    bool used_default_material = false;
    uint32_t default_material_index = 0;
};

bool walk_node(const Glb &g, int node_index, const M4 &trs, Gather &out, int depth) {
    const JValue *nodes = g.arr("nodes"), *meshes = g.arr("meshes");
    if (!nodes || node_index < 0 || (size_t)node_index >= nodes->arr.size() || depth > 256) return false;
    const JValue &node = *nodes->arr[node_index];

    M4 node_trs = identity();
    if (const JValue *mat = node.get("matrix"); mat && mat->arr.size() == 16) {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) node_trs.m[c][r] = (float)mat->arr[c * 4 + r]->num;
    } else {
        float t[3] = {0, 0, 0}, s[3] = {1, 1, 1};
        Quat q{0, 0, 0, 1};
        if (const JValue *v = node.get("translation"); v && v->arr.size() == 3)
            for (int i = 0; i < 3; ++i) t[i] = (float)v->arr[i]->num;
        if (const JValue *v = node.get("scale"); v && v->arr.size() == 3)
            for (int i = 0; i < 3; ++i) s[i] = (float)v->arr[i]->num;
        if (const JValue *v = node.get("rotation"); v && v->arr.size() == 4)
            q = Quat{(float)v->arr[0]->num, (float)v->arr[1]->num, (float)v->arr[2]->num, (float)v->arr[3]->num};
        node_trs = from_trs(t, q, s);
    }
    M4 new_trs = mul(trs, node_trs);
    float node_scale[3];
    Quat node_quat;
    decompose(new_trs, node_scale, node_quat);

    int mesh_index = (int)node.number_or("mesh", -1);
    if (mesh_index >= 0 && meshes && (size_t)mesh_index < meshes->arr.size()) {
        const JValue *prims = meshes->arr[mesh_index]->get("primitives");
        if (prims)
            for (auto &pp : prims->arr) {
                const JValue &prim = *pp;
                int mode = (int)prim.number_or("mode", 4);
                if (mode != 4) continue;   /* SortByPrimitiveType + assert_eq!(f.0.len(), 3): triangles only */
                const JValue *attrs = prim.get("attributes");
                if (!attrs) continue;
                std::vector<double> pos, nor, uv, idx;
                size_t n_pos = 0, n_nor = 0, n_uv = 0, n_idx = 0;
                if (!read_accessor(g, (int)attrs->number_or("POSITION", -1), 3, pos, n_pos)) return false;
                bool has_nor = read_accessor(g, (int)attrs->number_or("NORMAL", -1), 3, nor, n_nor);
                bool has_uv = read_accessor(g, (int)attrs->number_or("TEXCOORD_0", -1), 2, uv, n_uv);
                std::vector<double> tan;
                size_t n_tan = 0;
                bool has_tan = read_accessor(g, (int)attrs->number_or("TANGENT", -1), 4, tan, n_tan) && n_tan == n_pos;
                int idx_acc = (int)prim.number_or("indices", -1);
                if (idx_acc >= 0) {
                    if (!read_accessor(g, idx_acc, 1, idx, n_idx)) return false;
                } else {
                    n_idx = n_pos;
                    idx.resize(n_idx);
                    for (size_t i = 0; i < n_idx; ++i) idx[i] = (double)i;
                }
                uint32_t material;
                int mat_index = (int)prim.number_or("material", -1);
                if (mat_index >= 0) {
                    material = (uint32_t)mat_index;
                } else {
                    out.used_default_material = true;
                    material = out.default_material_index;
                }

                uint32_t triangle_offset = (uint32_t)out.vertices.size();
                for (size_t i = 0; i < n_pos; ++i) {
                    float v[3] = {(float)pos[3 * i], (float)pos[3 * i + 1], (float)pos[3 * i + 2]};
                    /* new_trs.mul_vec4((v, 1)): x*c0 + y*c1 + z*c2 + c3 */
                    float w[3];
                    for (int r = 0; r < 3; ++r)
                        w[r] = ((new_trs.m[0][r] * v[0] + new_trs.m[1][r] * v[1]) + new_trs.m[2][r] * v[2]) + new_trs.m[3][r];
                    out.vertices.push_back(Vec4f{w[0], w[2], w[1], 1.0f});
                }
                for (size_t k = 0; k < n_idx; ++k)       /* an index must name a vertex of THIS primitive (the reference would panic on it) */
                    if (!(idx[k] >= 0.0) || !(idx[k] < (double)n_pos)) return false;
                if (out.vertices.size() + n_pos > 0xfffffff0ull) return false;
                for (size_t f = 0; f + 2 < n_idx; f += 3) {
                    uint32_t i0 = (uint32_t)idx[f], i1 = (uint32_t)idx[f + 1], i2 = (uint32_t)idx[f + 2];
                    out.indices.push_back(rpt_triangle{triangle_offset + i0, triangle_offset + i2, triangle_offset + i1, material});
                }
                if (has_nor) {
                    for (size_t i = 0; i < n_nor; ++i) {
                        float n[3] = {(float)nor[3 * i] / node_scale[0], (float)nor[3 * i + 1] / node_scale[1],
                                      (float)nor[3 * i + 2] / node_scale[2]};
                        float r[3];
                        quat_rotate(node_quat, n, r);
                        float inv = 1.0f / std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                        out.normals.push_back(Vec4f{r[0] * inv, r[2] * inv, r[1] * inv, 0.0f});
                    }
                }
                /* tangents (asset.rs:111-114): the glTF importer hands a file's TANGENT attribute through (CalcTangentSpace
                 * leaves meshes that have tangents alone); without one assimp derives them from positions and uvs — see
                 * derive_tangents below */
                out.tangents.resize(triangle_offset, Vec4f{0, 0, 0, 0});
                if (has_tan) {
                    for (size_t i = 0; i < n_tan; ++i) {
                        float t[3] = {(float)tan[4 * i] / node_scale[0], (float)tan[4 * i + 1] / node_scale[1], (float)tan[4 * i + 2] / node_scale[2]};
                        float r[3];
                        quat_rotate(node_quat, t, r);
                        float inv = 1.0f / std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                        out.tangents.push_back(Vec4f{r[0] * inv, r[2] * inv, r[1] * inv, 0.0f});
                    }
                } else {
                    out.any_tangent_missing = true;
                    out.tangents.resize(out.vertices.size(), Vec4f{0, 0, 0, 0});
                }
                if (has_uv) {
                    for (size_t i = 0; i < n_uv; ++i) {
                        out.uvs.push_back((float)uv[2 * i]);
                        out.uvs.push_back(1.0f - (float)uv[2 * i + 1]);   /* assimp glTF2 importer flips v */
                    }
                } else {
                    out.uvs.resize(out.vertices.size() * 2, 0.0f);
                }
            }
    }
    if (const JValue *children = node.get("children"))
        for (auto &c : children->arr)
            if (!walk_node(g, (int)c->num, new_trs, out, depth + 1)) return false;
    return true;
}

}  // namespace

bool finish_world(World &w, std::vector<Vec4f> &vertices, std::vector<Vec4f> &normals, std::vector<Vec4f> &tangents,
                  std::vector<float> &uvs) {
    /* asset.rs:196 BVH (reorders indices), :201-202 light table, :206-215 packing */
    if (!build_world_bvh(vertices.data(), vertices.size(), w.indices, w.nodes)) return false;
    w.max_depth = bvh_max_depth(w.nodes);
    w.light_pick = build_light_pick_table(vertices.data(), w.indices.data(), w.indices.size(), w.materials.data(), &w.n_emissive);
    w.per_vertex.resize(vertices.size());
    for (size_t i = 0; i < vertices.size(); ++i) {
        rpt_per_vertex_data pv{};
        memcpy(pv.vertex, &vertices[i], 16);
        if (i < normals.size()) memcpy(pv.normal, &normals[i], 16);
        if (i < tangents.size()) memcpy(pv.tangent, &tangents[i], 16);
        if (2 * i + 1 < uvs.size()) { pv.uv0[0] = uvs[2 * i]; pv.uv0[1] = uvs[2 * i + 1]; }
        w.per_vertex[i] = pv;
    }
    return true;
}

namespace {

/* Vertices without a TANGENT attribute: per-triangle tangent from the uv gradient (the construction assimp's
 * CalcTangentsProcess starts from: T = (dv2 * e1 - dv1 * e2) / (du1 * dv2 - du2 * dv1)), accumulated per vertex,
 * made orthogonal to the normal and normalised.  assimp additionally merges across its own vertex joins, which cannot
 * be reproduced without it: parity for normal-mapped materials is defined at the buffer boundary (TANGENT in the file,
 * or buffers handed over directly). */
void derive_tangents(const std::vector<Vec4f> &v, const std::vector<Vec4f> &n, const std::vector<float> &uv,
                     const std::vector<rpt_triangle> &tris, std::vector<Vec4f> &tangents) {
    std::vector<Vec4f> acc(v.size(), Vec4f{0, 0, 0, 0});
    for (const rpt_triangle &t : tris) {
        const uint32_t i0 = t.v0, i1 = t.v1, i2 = t.v2;
        if (2 * (size_t)std::max(i0, std::max(i1, i2)) + 1 >= uv.size()) continue;
        float e1[3] = {v[i1].x - v[i0].x, v[i1].y - v[i0].y, v[i1].z - v[i0].z};
        float e2[3] = {v[i2].x - v[i0].x, v[i2].y - v[i0].y, v[i2].z - v[i0].z};
        float du1 = uv[2 * i1] - uv[2 * i0], dv1 = uv[2 * i1 + 1] - uv[2 * i0 + 1];
        float du2 = uv[2 * i2] - uv[2 * i0], dv2 = uv[2 * i2 + 1] - uv[2 * i0 + 1];
        float det = du1 * dv2 - du2 * dv1;
        if (det == 0.0f) continue;
        float r = 1.0f / det;
        float tx = (dv2 * e1[0] - dv1 * e2[0]) * r, ty = (dv2 * e1[1] - dv1 * e2[1]) * r, tz = (dv2 * e1[2] - dv1 * e2[2]) * r;
        for (uint32_t i : {i0, i1, i2}) { acc[i].x += tx; acc[i].y += ty; acc[i].z += tz; }
    }
    for (size_t i = 0; i < v.size(); ++i) {
        if (i < tangents.size() && (tangents[i].x != 0.0f || tangents[i].y != 0.0f || tangents[i].z != 0.0f)) continue;   /* from the file */
        float t[3] = {acc[i].x, acc[i].y, acc[i].z};
        if (i < n.size()) {
            float d = t[0] * n[i].x + t[1] * n[i].y + t[2] * n[i].z;
            t[0] -= d * n[i].x; t[1] -= d * n[i].y; t[2] -= d * n[i].z;
        }
        float len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        if (tangents.size() <= i) tangents.resize(i + 1, Vec4f{0, 0, 0, 0});
        tangents[i] = len > 0.0f ? Vec4f{t[0] / len, t[1] / len, t[2] / len, 0.0f} : Vec4f{0, 0, 0, 0};
    }
}

/* images[textures[index].source] of the GLB, decoded (asset.rs:27-44).  index < 0: no texture. */
bool load_gltf_texture(const Glb &g, const JValue *texinfo, Image8 &img, bool &present) {
    present = false;
    if (!texinfo) return true;
    int ti = (int)texinfo->number_or("index", -1);
    const JValue *textures = g.arr("textures"), *images = g.arr("images"), *views = g.arr("bufferViews");
    if (ti < 0 || !textures || (size_t)ti >= textures->arr.size()) return true;
    int src = (int)textures->arr[ti]->number_or("source", -1);
    if (src < 0 || !images || (size_t)src >= images->arr.size()) return true;
    const JValue &im = *images->arr[src];
    int bv = (int)im.number_or("bufferView", -1);
    if (bv < 0 || !views || (size_t)bv >= views->arr.size()) { set_error("GLB image without an embedded bufferView (external uri) is not supported"); return false; }
    double off = views->arr[bv]->number_or("byteOffset", 0), len = views->arr[bv]->number_or("byteLength", 0);
    if (!(off >= 0) || !(len >= 8) || off + len > (double)g.bin.size()) { set_error("GLB image bufferView out of range"); return false; }
    const uint8_t *p = g.bin.data() + (size_t)off;
    /* PNG or JPEG (the two formats glTF allows).  An image that does not decode (here: arithmetic-coded, 12-bit, CMYK ... JPEG) is an
     * ABSENT texture, as the reference's `.decode().ok()?` makes it (src/asset.rs:35) — not a failed scene load */
    if (!decode_image(p, (size_t)len, img)) return true;
    present = true;
    return true;
}

}  // namespace

bool load_glb(const char *path, World &out, uint32_t flags) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return false; }
    std::vector<uint8_t> data;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 20) { fclose(f); set_error("file too small for GLB"); return false; }
    data.resize((size_t)sz);
    size_t got = fread(data.data(), 1, data.size(), f);
    fclose(f);
    if (got != data.size()) { set_error("short read"); return false; }
    uint32_t magic, version, total;
    memcpy(&magic, &data[0], 4); memcpy(&version, &data[4], 4); memcpy(&total, &data[8], 4);
    if (magic != 0x46546C67u || version != 2) { set_error("not a glTF 2.0 binary"); return false; }

    Glb g;
    size_t at = 12;
    while (at + 8 <= data.size()) {
        uint32_t clen, ctype;
        memcpy(&clen, &data[at], 4); memcpy(&ctype, &data[at + 4], 4);
        at += 8;
        if (at + clen > data.size()) { set_error("truncated GLB chunk"); return false; }
        if (ctype == 0x4E4F534Au) {
            JParser p{(const char *)&data[at], (const char *)&data[at] + clen};
            g.root = p.parse();
            if (!p.ok || g.root->kind != JValue::Obj) { set_error("GLB JSON parse error"); return false; }
        } else if (ctype == 0x004E4942u && g.bin.empty()) {
            g.bin.assign(data.begin() + (long)at, data.begin() + (long)(at + clen));
        }
        at += clen;
    }
    if (!g.root) { set_error("GLB has no JSON chunk"); return false; }

    const JValue *materials = g.arr("materials");
    size_t n_file_materials = materials ? materials->arr.size() : 0;
    Gather ga;
    ga.default_material_index = (uint32_t)n_file_materials;

    const JValue *scenes = g.arr("scenes");
    int scene_index = (int)g.root->number_or("scene", 0);
    if (!scenes || scenes->arr.empty()) { set_error("GLB has no scenes"); return false; }
    if ((size_t)scene_index >= scenes->arr.size()) scene_index = 0;
    const JValue *roots = scenes->arr[scene_index]->get("nodes");
    if (roots)
        for (auto &r : roots->arr)
            if (!walk_node(g, (int)r->num, identity(), ga, 0)) { set_error("GLB node/mesh data invalid"); return false; }
    if (ga.indices.empty()) { set_error("GLB contains no triangles"); return false; }

    /* materials (asset.rs:135-175) */
    out.materials.assign(n_file_materials + (ga.used_default_material ? 1 : 0), rpt_material_data{});
    std::vector<Image8> textures;          /* in the reference's push order: per material albedo, metallic, roughness, normals */
    for (size_t i = 0; i < out.materials.size(); ++i) {
        rpt_material_data &m = out.materials[i];
        float base[4] = {1, 1, 1, 1}, emissive[3] = {0, 0, 0};
        float metallic = 1.0f, roughness = 1.0f;
        float strength = 1.0f;
        bool has_strength = false;
        if (i < n_file_materials) {
            const JValue &jm = *materials->arr[i];
            if (const JValue *pbr = jm.get("pbrMetallicRoughness")) {
                if (const JValue *c = pbr->get("baseColorFactor"); c && c->arr.size() == 4)
                    for (int k = 0; k < 4; ++k) base[k] = (float)c->arr[k]->num;
                metallic = (float)pbr->number_or("metallicFactor", 1.0);
                roughness = (float)pbr->number_or("roughnessFactor", 1.0);
            }
            if (const JValue *e = jm.get("emissiveFactor"); e && e->arr.size() == 3)
                for (int k = 0; k < 3; ++k) emissive[k] = (float)e->arr[k]->num;
            if (flags & RPT_LOAD_EMISSIVE_STRENGTH)
                if (const JValue *ext = jm.get("extensions"))
                    if (const JValue *es = ext->get("KHR_materials_emissive_strength")) {
                        strength = (float)es->number_or("emissiveStrength", 1.0);
                        has_strength = true;
                    }
            /* textures (asset.rs:140-160).  assimp's glTF2 importer exposes metallicRoughnessTexture under BOTH
             * aiTextureType_METALNESS and aiTextureType_DIFFUSE_ROUGHNESS, so the reference atlases it twice. */
            const JValue *pbr = jm.get("pbrMetallicRoughness");
            Image8 img;
            bool present = false;
            if (!load_gltf_texture(g, pbr ? pbr->get("baseColorTexture") : nullptr, img, present)) return false;
            if (present) { albedo_gamma_to_linear(img); textures.push_back(img); m.has_albedo_texture = 1; }
            if (!load_gltf_texture(g, pbr ? pbr->get("metallicRoughnessTexture") : nullptr, img, present)) return false;
            if (present) { textures.push_back(img); m.has_metallic_texture = 1; textures.push_back(img); m.has_roughness_texture = 1; }
            if (!load_gltf_texture(g, jm.get("normalTexture"), img, present)) return false;
            if (present) { textures.push_back(img); m.has_normal_texture = 1; }
        }
        for (int k = 0; k < 4; ++k) m.albedo[k] = base[k];
        /* asset.rs:163-166: "Multiply by 15 since assimp 5.2.5 doesn't support emissive strength" — the reference's
         * behaviour and the default here; RPT_LOAD_EMISSIVE_STRENGTH honours KHR_materials_emissive_strength instead */
        const float gain = has_strength ? strength : 15.0f;
        m.emissive[0] = emissive[0] * gain; m.emissive[1] = emissive[1] * gain;
        m.emissive[2] = emissive[2] * gain; m.emissive[3] = 1.0f * gain;
        for (int k = 0; k < 4; ++k) { m.metallic[k] = metallic; m.roughness[k] = roughness; }
    }
    for (const rpt_triangle &t : ga.indices)
        if (t.material >= out.materials.size()) { set_error("material index out of range"); return false; }

    if (!textures.empty()) {
        /* atlas.rs:26-90 + asset.rs:174-187: pack, then hand the uvst out in push order */
        std::vector<std::array<float, 4>> sts;
        out.atlas_w = out.atlas_h = 4096;
        pack_textures(textures, out.atlas_w, out.atlas_h, out.atlas, sts);
        size_t at = 0;
        for (rpt_material_data &m : out.materials) {
            if (m.has_albedo_texture) memcpy(m.albedo, sts[at++].data(), 16);
            if (m.has_metallic_texture) memcpy(m.metallic, sts[at++].data(), 16);
            if (m.has_roughness_texture) memcpy(m.roughness, sts[at++].data(), 16);
            if (m.has_normal_texture) memcpy(m.normals, sts[at++].data(), 16);
        }
        if (ga.any_tangent_missing) derive_tangents(ga.vertices, ga.normals, ga.uvs, ga.indices, ga.tangents);
    }
    out.indices = ga.indices;
    return finish_world(out, ga.vertices, ga.normals, ga.tangents, ga.uvs);
}

}  // namespace rpth
