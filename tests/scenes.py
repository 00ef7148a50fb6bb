"""Procedural test scenes (data built in Python, run through the reference's BVH / light-table / packing steps)."""
import importlib

import numpy as np


def textured_scene(seed=7):
    """Floor, back wall, a tilted plate with uvs outside [0,1] (wrap), a sphere-ish blob and an area light.
    Materials exercise every texture flag (albedo / metallic / roughness / normal map) of get_pbr_bsdf
    (reference: kernels/src/bsdf.rs:354-387, lib.rs:132-141) through a small RGBA8 atlas."""
    rpt = importlib.import_module("rust-path-tracer_amd")
    rng = np.random.default_rng(seed)
    verts, normals, uvs, tris = [], [], [], []

    def quad(p0, p1, p2, p3, n, mat, uv_scale=1.0, uv_off=0.0):
        b = len(verts)
        verts.extend([p0, p1, p2, p3])
        normals.extend([n] * 4)
        uvs.extend([[uv_off, uv_off], [uv_off + uv_scale, uv_off], [uv_off + uv_scale, uv_off + uv_scale], [uv_off, uv_off + uv_scale]])
        tris.extend([[b, b + 1, b + 2, mat], [b, b + 2, b + 3, mat]])

    quad([-4, 0, -2], [4, 0, -2], [4, 0, 6], [-4, 0, 6], [0, 1, 0], 0)                       # floor
    quad([-4, 0, 6], [4, 0, 6], [4, 5, 6], [-4, 5, 6], [0, 0, -1], 1)                        # back wall
    quad([-2.5, 0.3, 1.5], [-0.5, 0.8, 2.5], [-0.5, 2.2, 2.0], [-2.5, 1.7, 1.0], [0.3, 0.3, -0.9], 2, uv_scale=2.7, uv_off=-0.8)
    quad([0.8, 0.2, 1.0], [2.6, 0.2, 1.8], [2.6, 1.9, 1.8], [0.8, 1.9, 1.0], [0.4, 0.0, -0.9], 3)
    quad([-1, 4.2, 1], [1, 4.2, 1], [1, 4.2, 3], [-1, 4.2, 3], [0, -1, 0], 4)               # light, facing down
    # a coarse blob so the BVH has some depth
    for _ in range(60):
        c = np.array([rng.uniform(-3, 3), rng.uniform(0.2, 2.5), rng.uniform(2.5, 5.5)])
        d = rng.normal(size=(3, 3)) * 0.25
        b = len(verts)
        n = np.cross(d[1] - d[0], d[2] - d[0])
        n = n / np.linalg.norm(n)
        for k in range(3):
            verts.append(list(c + d[k]))
            normals.append(list(n))
            uvs.append([rng.uniform(0, 1), rng.uniform(0, 1)])
        tris.append([b, b + 1, b + 2, int(rng.integers(0, 4))])

    m = np.zeros(5, rpt._ffi.MATERIAL_DTYPE)
    m["albedo"] = [0.8, 0.8, 0.8, 1.0]
    m["roughness"] = 0.6
    m["metallic"] = 0.1
    # atlas rectangles (u0, v0, su, sv) inside a 64x64 atlas: four 32x32 quadrants
    q = {"a": [0.0, 0.0, 0.5, 0.5], "b": [0.5, 0.0, 0.5, 0.5], "c": [0.0, 0.5, 0.5, 0.5], "d": [0.5, 0.5, 0.5, 0.5]}
    m["albedo"][0] = q["a"]; m["has_albedo_texture"][0] = 1
    m["normals"][0] = q["d"]; m["has_normal_texture"][0] = 1
    m["roughness"][0] = q["c"]; m["has_roughness_texture"][0] = 1
    m["metallic"][1] = q["b"]; m["has_metallic_texture"][1] = 1
    m["albedo"][2] = q["a"]; m["has_albedo_texture"][2] = 1
    m["roughness"][2] = 0.15; m["metallic"][2] = 0.9
    m["normals"][3] = q["d"]; m["has_normal_texture"][3] = 1
    m["emissive"][4] = [12.0, 11.0, 9.0, 15.0]
    w = rpt.World.from_buffers(np.array(verts, np.float32), np.array(normals, np.float32), np.array(uvs, np.float32),
                               np.array(tris, np.uint32), m)
    # tangents (only read when a normal texture exists): any unit vector not parallel to the normal
    n = w.per_vertex["normal"][:, :3]
    t = np.cross(n, np.array([0.0, 0.0, 1.0], np.float32))
    bad = np.linalg.norm(t, axis=1) < 1e-3
    t[bad] = np.cross(n[bad], np.array([1.0, 0.0, 0.0], np.float32))
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    w.per_vertex["tangent"][:, :3] = t.astype(np.float32)
    atlas = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    atlas[32:, 32:, :2] = rng.integers(96, 160, (32, 32, 2), dtype=np.uint8)     # normal-map quadrant: near +z
    atlas[32:, 32:, 2] = 255
    w.atlas = atlas
    skybox = rng.uniform(0.0, 2.0, (8, 16, 4)).astype(np.float32)
    skybox[..., 3] = 1.0
    return w, skybox


def pbrtest_textured_scene():
    """BASELINE config[3] as it is written — "PBRTest.glb with albedo/normal/rough/metal textures" — on the shipped file, which carries
    NO texture: the file's own buffers (vertices, uvs, BVH, light table) + SYNTHETIC textures, labelled as such.
    Every one of the 26 materials gets the four maps get_pbr_bsdf / the normal-map branch read (bsdf.rs:354-387, lib.rs:132-141), in the
    order World::from_path pushes them (albedo, metallic, roughness, normals: src/asset.rs:138-160), each in the rectangle of the
    4096 x 4096 RGBA8 atlas the reference's own packer gives 104 textures (pack_rects = src/atlas.rs:26-71; uvst = atlas.rs:16-23):
    40 rectangles of 512^2 and 64 of 256^2 texels.  Contents: a two-colour checker with a gradient inside each square (albedo), a rippled
    tangent-space normal, roughness / metallic ramps around the material's own factor.  PBRTest.glb has no TANGENT attribute (the
    reference would get assimp's CalculateTangentSpace, asset.rs:65): tangents are a unit vector perpendicular to the vertex normal."""
    rpt = importlib.import_module("rust-path-tracer_amd")
    w = rpt.World.from_path(rpt.fixture("PBRTest.glb"))
    A = 4096
    n_mat = len(w.materials)
    rects = pack_rects(4 * n_mat, A)
    atlas = np.zeros((A, A, 4), np.uint8)
    atlas[..., 3] = 255
    m = w.materials.copy()

    def uvst(r):
        x, y, rw, rh = r
        return [np.float32(x) / np.float32(A), np.float32(y) / np.float32(A), np.float32(rw) / np.float32(A), np.float32(rh) / np.float32(A)]

    def grid(rw, rh):
        yy, xx = np.mgrid[0:rh, 0:rw].astype(np.float32)
        return xx, yy, xx / np.float32(rw), yy / np.float32(rh)

    def put(r, rgb):
        x, y, rw, rh = r
        atlas[y:y + rh, x:x + rw, :3] = np.clip(np.rint(rgb * 255.0), 0, 255).astype(np.uint8)

    for i in range(n_mat):
        base_rough, base_metal = float(m["roughness"][i][0]), float(m["metallic"][i][0])
        r_alb, r_met, r_rough, r_nrm = rects[4 * i: 4 * i + 4]
        # albedo: checker of 32-texel squares between two colours of the material's own hue, a gradient inside every square
        xx, yy, u, v = grid(r_alb[2], r_alb[3])
        hue = (i * 0.61803398875) % 1.0
        c0 = np.array([0.5 + 0.5 * np.cos(2 * np.pi * (hue + k / 3.0)) for k in range(3)], np.float32) * 0.7 + 0.15
        c1 = 1.0 - 0.6 * c0
        check = ((xx // 32 + yy // 32) % 2)[..., None]
        shade = (0.75 + 0.25 * ((xx % 32) / 32.0))[..., None]
        put(r_alb, (check * c0 + (1 - check) * c1) * shade)
        m["albedo"][i] = uvst(r_alb); m["has_albedo_texture"][i] = 1
        # metallic: vertical ramp around the factor; roughness: horizontal ramp around the factor (the kernels read .x)
        xx, yy, u, v = grid(r_met[2], r_met[3])
        put(r_met, np.repeat(np.clip(base_metal + 0.5 * (v - 0.5), 0.0, 1.0)[..., None], 3, 2))
        m["metallic"][i] = uvst(r_met); m["has_metallic_texture"][i] = 1
        xx, yy, u, v = grid(r_rough[2], r_rough[3])
        put(r_rough, np.repeat(np.clip(base_rough + 0.4 * (u - 0.5), 0.03, 1.0)[..., None], 3, 2))
        m["roughness"][i] = uvst(r_rough); m["has_roughness_texture"][i] = 1
        # normal map: ripples of 64 texels, at most ~17 degrees off +z
        xx, yy, u, v = grid(r_nrm[2], r_nrm[3])
        nx = 0.3 * np.sin(2 * np.pi * xx / 64.0)
        ny = 0.3 * np.cos(2 * np.pi * yy / 48.0)
        nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
        put(r_nrm, np.stack([nx, ny, nz], -1) * 0.5 + 0.5)
        m["normals"][i] = uvst(r_nrm); m["has_normal_texture"][i] = 1
    w.materials = m
    n = w.per_vertex["normal"][:, :3]
    t = np.cross(n, np.array([0.0, 1.0, 0.0], np.float32))
    bad = np.linalg.norm(t, axis=1) < 1e-3
    t[bad] = np.cross(n[bad], np.array([1.0, 0.0, 0.0], np.float32))
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    w.per_vertex["tangent"][:, :3] = t.astype(np.float32)
    w.atlas = atlas
    return w


def deep_bvh_scene(n_triangles=200_000, seed=1):
    """Stand-in for the missing BreakTime.glb (BASELINE config 5): clustered long thin triangles
    inside a closed room, a few emissive panels.  Long thin primitives overlap heavily, which makes the binned-SAH
    BVH deep and the traversal divergent."""
    rpt = importlib.import_module("rust-path-tracer_amd")
    rng = np.random.default_rng(seed)
    n_clusters = 64
    centers = np.stack([rng.uniform(-3, 3, n_clusters), rng.uniform(0.3, 4.0, n_clusters), rng.uniform(0.5, 7.0, n_clusters)], 1)
    per = n_triangles // n_clusters
    verts, tris = [], []
    for c in centers:
        base = c + rng.normal(size=(per, 3)) * rng.uniform(0.15, 0.6)
        axis = rng.normal(size=(per, 3))
        axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        side = np.cross(axis, rng.normal(size=(per, 3)))
        side /= np.linalg.norm(side, axis=1, keepdims=True)
        length = rng.uniform(0.01, 0.12, (per, 1))
        width = rng.uniform(0.002, 0.02, (per, 1))
        v0 = base - axis * length
        v1 = base + axis * length
        v2 = base + side * width
        b = len(verts) * 3
        verts.append(np.stack([v0, v1, v2], 1).reshape(-1, 3))
        idx = b + np.arange(per * 3).reshape(per, 3)
        tris.append(np.concatenate([idx, rng.integers(0, 3, (per, 1))], 1))
    verts = np.concatenate(verts).astype(np.float32)
    tris = np.concatenate(tris).astype(np.uint32)
    # flat normals for the cluster triangles (each owns its 3 vertices)
    p = verts[tris[:, :3].astype(np.int64)]
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    normals = np.repeat(fn, 3, axis=0).astype(np.float32)
    # closed room (every wall has its own 4 vertices and an INWARD normal, so bounces stay inside) + light panel
    corners = np.array([[-5, 0, -1], [5, 0, -1], [5, 0, 9], [-5, 0, 9], [-5, 6, -1], [5, 6, -1], [5, 6, 9], [-5, 6, 9]], np.float32)
    walls = [((0, 1, 2, 3), (0, 1, 0)), ((7, 6, 5, 4), (0, -1, 0)), ((0, 4, 5, 1), (0, 0, 1)), ((3, 2, 6, 7), (0, 0, -1)),
             ((0, 3, 7, 4), (1, 0, 0)), ((1, 5, 6, 2), (-1, 0, 0))]
    extra_v, extra_n, extra_t = [], [], []
    b = len(verts)
    for q, nrm in walls:
        k = b + len(extra_v)
        extra_v += [corners[i] for i in q]
        extra_n += [nrm] * 4
        extra_t += [[k, k + 1, k + 2, 3], [k, k + 2, k + 3, 3]]
    k = b + len(extra_v)
    extra_v += [[-1.5, 5.9, 2], [1.5, 5.9, 2], [1.5, 5.9, 5], [-1.5, 5.9, 5]]
    extra_n += [(0, -1, 0)] * 4
    extra_t += [[k, k + 2, k + 1, 4], [k, k + 3, k + 2, 4]]        # wound so the geometric normal faces down (emissives are single sided)
    verts = np.concatenate([verts, np.array(extra_v, np.float32)])
    normals = np.concatenate([normals, np.array(extra_n, np.float32)])
    tris = np.concatenate([tris, np.array(extra_t, np.uint32)])
    m = np.zeros(5, rpt._ffi.MATERIAL_DTYPE)
    m["albedo"][:] = [[0.7, 0.2, 0.2, 1], [0.2, 0.7, 0.2, 1], [0.3, 0.3, 0.8, 1], [0.75, 0.75, 0.75, 1], [0, 0, 0, 1]]
    m["roughness"][:, :] = np.array([0.3, 0.6, 0.9, 1.0, 1.0], np.float32)[:, None]
    m["metallic"][:, :] = np.array([0.8, 0.0, 0.3, 0.0, 0.0], np.float32)[:, None]
    m["emissive"][4] = [18.0, 17.0, 15.0, 15.0]
    return rpt.World.from_buffers(verts, normals.astype(np.float32), None, tris, m)


def scatter_scene(n_triangles=60_000, seed=9):
    """Small triangles scattered uniformly in a closed room with a light: the builder splits down to leaves of one or two
    triangles, so the BVH has MORE THAN 65 536 nodes — 32-bit stack entries in the global-memory walks, which no shipped
    scene reaches (PBRTest: 47 637 nodes)."""
    rpt = importlib.import_module("rust-path-tracer_amd")
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(-2.8, 2.8, n_triangles), rng.uniform(0.2, 3.6, n_triangles), rng.uniform(-0.6, 4.6, n_triangles)], 1)
    a = rng.normal(size=(n_triangles, 3)) * 0.03
    b = rng.normal(size=(n_triangles, 3)) * 0.03
    verts = np.stack([c - a, c + a, c + b], 1).reshape(-1, 3).astype(np.float32)
    tris = np.concatenate([np.arange(3 * n_triangles).reshape(-1, 3), rng.integers(0, 2, (n_triangles, 1))], 1).astype(np.uint32)
    p = verts.reshape(-1, 3, 3)
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    normals = np.repeat(fn, 3, axis=0).astype(np.float32)
    corners = np.array([[-3, 0, -1], [3, 0, -1], [3, 0, 5], [-3, 0, 5], [-3, 4, -1], [3, 4, -1], [3, 4, 5], [-3, 4, 5]], np.float32)
    walls = [((0, 1, 2, 3), (0, 1, 0)), ((7, 6, 5, 4), (0, -1, 0)), ((3, 2, 6, 7), (0, 0, -1)), ((0, 3, 7, 4), (1, 0, 0)), ((1, 5, 6, 2), (-1, 0, 0))]
    ev, en, et = [], [], []
    b0 = len(verts)
    for q, nrm in walls:
        k = b0 + len(ev)
        ev += [corners[i] for i in q]
        en += [nrm] * 4
        et += [[k, k + 1, k + 2, 2], [k, k + 2, k + 3, 2]]
    k = b0 + len(ev)
    ev += [[-1, 3.9, 1], [1, 3.9, 1], [1, 3.9, 3], [-1, 3.9, 3]]
    en += [(0, -1, 0)] * 4
    et += [[k, k + 2, k + 1, 3], [k, k + 3, k + 2, 3]]
    verts = np.concatenate([verts, np.array(ev, np.float32)])
    normals = np.concatenate([normals, np.array(en, np.float32)])
    tris = np.concatenate([tris, np.array(et, np.uint32)])
    m = np.zeros(4, rpt._ffi.MATERIAL_DTYPE)
    m["albedo"][:] = [[0.8, 0.3, 0.2, 1], [0.2, 0.5, 0.8, 1], [0.75, 0.75, 0.75, 1], [0, 0, 0, 1]]
    m["roughness"][:, :] = np.array([0.4, 0.8, 1.0, 1.0], np.float32)[:, None]
    m["metallic"][:, :] = np.array([0.6, 0.0, 0.0, 0.0], np.float32)[:, None]
    m["emissive"][3] = [18.0, 17.0, 15.0, 15.0]
    return rpt.World.from_buffers(verts, normals.astype(np.float32), None, tris, m)


def fat_leaf_scene(n_stack=300, seed=3):
    """A leaf the builder cannot split: n_stack triangles with ONE centroid (rotated copies about it) inside a small room
    with a light.  Every binned-SAH split leaves one side empty, so they stay one leaf of >= 128 triangles: no LDS image
    (leaves of 64+ triangles), several 64-lane rounds of the wave-cooperative leaf test."""
    rpt = importlib.import_module("rust-path-tracer_amd")
    rng = np.random.default_rng(seed)
    c = np.array([0.0, 1.2, 1.5])
    ang = rng.uniform(0, 2 * np.pi, n_stack)
    tilt = rng.uniform(-0.6, 0.6, n_stack)
    r = rng.uniform(0.3, 0.9, n_stack)
    verts, tris = [], []
    for a, t, rad in zip(ang, tilt, r):
        u = np.array([np.cos(a), np.sin(t), np.sin(a)])
        u /= np.linalg.norm(u)
        v = np.cross(u, [0.3, 1.0, 0.2])
        v /= np.linalg.norm(v)
        p0, p1 = c + rad * u, c + rad * (-0.5 * u + 0.866 * v)
        p2 = 3.0 * c - p0 - p1                                    # the third corner makes the centroid exactly (up to rounding) c
        k = len(verts)
        verts += [p0, p1, p2]
        tris.append([k, k + 1, k + 2, int(rng.integers(0, 2))])
    verts = np.array(verts, np.float32)
    p = verts.reshape(-1, 3, 3)
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    normals = np.repeat(fn, 3, axis=0).astype(np.float32)
    corners = np.array([[-3, 0, -1], [3, 0, -1], [3, 0, 5], [-3, 0, 5], [-3, 4, -1], [3, 4, -1], [3, 4, 5], [-3, 4, 5]], np.float32)
    walls = [((0, 1, 2, 3), (0, 1, 0)), ((7, 6, 5, 4), (0, -1, 0)), ((3, 2, 6, 7), (0, 0, -1)), ((0, 3, 7, 4), (1, 0, 0)), ((1, 5, 6, 2), (-1, 0, 0))]
    ev, en, et = [], [], []
    b = len(verts)
    for q, nrm in walls:
        k = b + len(ev)
        ev += [corners[i] for i in q]
        en += [nrm] * 4
        et += [[k, k + 1, k + 2, 2], [k, k + 2, k + 3, 2]]
    k = b + len(ev)
    ev += [[-1, 3.9, 1], [1, 3.9, 1], [1, 3.9, 3], [-1, 3.9, 3]]
    en += [(0, -1, 0)] * 4
    et += [[k, k + 2, k + 1, 3], [k, k + 3, k + 2, 3]]
    verts = np.concatenate([verts, np.array(ev, np.float32)])
    normals = np.concatenate([normals, np.array(en, np.float32)])
    tris = np.concatenate([np.array(tris, np.uint32), np.array(et, np.uint32)])
    m = np.zeros(4, rpt._ffi.MATERIAL_DTYPE)
    m["albedo"][:] = [[0.8, 0.3, 0.2, 1], [0.2, 0.5, 0.8, 1], [0.75, 0.75, 0.75, 1], [0, 0, 0, 1]]
    m["roughness"][:, :] = np.array([0.4, 0.8, 1.0, 1.0], np.float32)[:, None]
    m["metallic"][:, :] = np.array([0.6, 0.0, 0.0, 0.0], np.float32)[:, None]
    m["emissive"][3] = [18.0, 17.0, 15.0, 15.0]
    return rpt.World.from_buffers(verts, normals.astype(np.float32), None, tris, m)


def foreign_pool(w):
    """The same tree as another builder might lay it out: an unused node at index 1, so every child pair sits at (even, odd) — not the (2p + 1, 2p + 2)
    pairs of the reference's builder (src/bvh.rs:296-320: node_count starts at 1).  A valid node buffer of the boundary (children adjacent, links in range)."""
    import copy
    out = copy.copy(w)
    nodes = w.nodes
    new = np.zeros(len(nodes) + 1, nodes.dtype)
    new[0] = nodes[0]
    new[2:] = nodes[1:]
    new[1] = nodes[len(nodes) - 1]                        # (never reached: a copy of some leaf)
    inner = new["triangle_count"] == 0
    inner[1] = False
    new["left_or_first"][inner] += 1
    out.nodes = new
    return out


def write_glb(path, positions, indices, *, normals=None, uvs=None, materials=None, images=None, textures=None, accessor_patch=None,
              node_extra=None):
    """Minimal glTF 2.0 binary writer for tests: one mesh, one primitive, optional embedded PNG images.
    positions (n, 3) float32, indices (m,) uint32.  `accessor_patch(accessors, views)` may corrupt the JSON on purpose."""
    import json
    import struct
    import numpy as np
    blob = bytearray()
    views, accessors = [], []

    def add(data, target=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob.extend(data)
        return len(views) - 1

    positions = np.ascontiguousarray(positions, np.float32)
    indices = np.ascontiguousarray(indices, np.uint32)
    attrs = {}
    accessors.append({"bufferView": add(positions.tobytes()), "componentType": 5126, "count": len(positions), "type": "VEC3"})
    attrs["POSITION"] = 0
    if normals is not None:
        accessors.append({"bufferView": add(np.ascontiguousarray(normals, np.float32).tobytes()), "componentType": 5126,
                          "count": len(normals), "type": "VEC3"})
        attrs["NORMAL"] = len(accessors) - 1
    if uvs is not None:
        accessors.append({"bufferView": add(np.ascontiguousarray(uvs, np.float32).tobytes()), "componentType": 5126,
                          "count": len(uvs), "type": "VEC2"})
        attrs["TEXCOORD_0"] = len(accessors) - 1
    accessors.append({"bufferView": add(indices.tobytes()), "componentType": 5125, "count": len(indices), "type": "SCALAR"})
    prim = {"attributes": attrs, "indices": len(accessors) - 1}
    if materials:
        prim["material"] = 0
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [dict({"mesh": 0}, **(node_extra or {}))],
           "meshes": [{"primitives": [prim]}], "accessors": accessors, "bufferViews": views}
    if materials:
        doc["materials"] = materials
    if images:
        doc["images"] = [{"bufferView": add(png), "mimeType": "image/png"} for png in images]
        doc["textures"] = textures if textures is not None else [{"source": i} for i in range(len(images))]
    if accessor_patch:
        accessor_patch(accessors, views)
    doc["buffers"] = [{"byteLength": len(blob)}]
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    while len(blob) % 4:
        blob.append(0)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + bytes(blob))
    return path


def png_bytes(rgb):
    """Encode an (H, W, 3|4) uint8 array as a PNG (filter 0, zlib level 1) — test input for the loader's decoder."""
    import struct
    import zlib
    import numpy as np
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, c = rgb.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb.reshape(h, w * c)], axis=1).tobytes()

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


def pack_rects(n_textures, atlas=4096):
    """src/atlas.rs:26-71 restated: the leaf rectangle (x, y, w, h) of every texture, in texture order."""
    from collections import deque
    q = deque([(0, 0, atlas, atlas)])
    while len(q) <= n_textures:
        x, y, w, h = q.popleft()
        hw, hh = w // 2, h // 2
        q.extend([(x, y, hw, hh), (x + hw, y, hw, hh), (x, y + hh, hw, hh), (x + hw, y + hh, hw, hh)])
    leafs = sorted(q, key=lambda r: -r[2])          # stable, by width descending
    return leafs[:n_textures]
