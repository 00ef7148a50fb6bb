"""ctypes wrapper of oracle/liboracle*.so — TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (as the checker / reported baseline, never as the product).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OracleScene(C.Structure):
    _fields_ = [
        ("per_vertex", C.c_void_p), ("n_vertices", C.c_size_t),
        ("indices", C.c_void_p), ("n_triangles", C.c_size_t),
        ("nodes", C.c_void_p), ("n_nodes", C.c_size_t),
        ("materials", C.c_void_p), ("n_materials", C.c_size_t),
        ("light_pick", C.c_void_p), ("n_light_pick", C.c_size_t),
        ("atlas", C.c_void_p), ("atlas_w", C.c_uint32), ("atlas_h", C.c_uint32),
        ("skybox", C.c_void_p), ("sky_w", C.c_uint32), ("sky_h", C.c_uint32),
    ]


class OracleStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("sky_evals", C.c_uint64), ("light_index_clamped", C.c_uint64),
        ("node_pops", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
        ("max_stack", C.c_uint32), ("error_flags", C.c_uint32),
        ("seconds", C.c_double), ("threads", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Oracle:
    """CPU restatement of trace_cpu / trace_pixel (reference: src/trace.rs:226-327, kernels/src/lib.rs:21-186)."""

    def __init__(self, backend="rpt_math"):
        name = "liboracle.so" if backend == "rpt_math" else "liboracle_libm.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make oracle`")
        self.lib = C.CDLL(path)
        self.lib.oracle_math_backend.restype = C.c_char_p
        self.lib.oracle_lds.restype = C.c_float
        assert self.lib.oracle_math_backend().decode() == backend

    def scene(self, world, atlas_f32=None, skybox_f32=None):
        """world: any object with per_vertex/indices/nodes/materials/light_pick numpy arrays."""
        s = OracleScene()
        keep = [world.per_vertex, world.indices, world.nodes, world.materials, world.light_pick]
        s.per_vertex, s.n_vertices = _p(world.per_vertex), len(world.per_vertex)
        s.indices, s.n_triangles = _p(world.indices), len(world.indices)
        s.nodes, s.n_nodes = _p(world.nodes), len(world.nodes)
        s.materials, s.n_materials = _p(world.materials), len(world.materials)
        s.light_pick, s.n_light_pick = _p(world.light_pick), len(world.light_pick)
        if atlas_f32 is None and getattr(world, "atlas", None) is not None:
            # CPU atlas texel = (r, g, b, 255) / 255 (reference: src/asset.rs:266-273)
            a = world.atlas.astype(np.float32)
            a[..., 3] = 255.0
            atlas_f32 = np.ascontiguousarray(a / np.float32(255.0), np.float32)
        if atlas_f32 is not None:
            atlas_f32 = np.ascontiguousarray(atlas_f32, np.float32)
            s.atlas, s.atlas_h, s.atlas_w = _p(atlas_f32), atlas_f32.shape[0], atlas_f32.shape[1]
            keep.append(atlas_f32)
        if skybox_f32 is not None:
            skybox_f32 = np.ascontiguousarray(skybox_f32, np.float32)
            s.skybox, s.sky_h, s.sky_w = _p(skybox_f32), skybox_f32.shape[0], skybox_f32.shape[1]
            keep.append(skybox_f32)
        s._keep = keep
        return s

    def trace_cpu(self, config, scene, rng, n_samples, accum=None, rect=None, threads=0):
        """Returns (accum HxWx4 float32 SUM, rng_after, stats). rng is consumed/updated in place on a copy."""
        W, H = config.width, config.height
        rng = np.ascontiguousarray(rng).copy()
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        else:
            accum = np.ascontiguousarray(accum, np.float32).copy()
        x0, y0, x1, y1 = rect if rect else (0, 0, W, H)
        st = OracleStats()
        rc = self.lib.oracle_trace_cpu(C.byref(config), C.byref(scene), _p(rng), _p(accum), C.c_uint32(n_samples),
                                       C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(x1), C.c_uint32(y1),
                                       C.c_int(threads), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"oracle_trace_cpu failed: {rc}")
        return accum, rng, st

    def dead_shadow_rays(self, config, scene, rng, n_samples, stride=1):
        """Analysis hook oracle_dead_shadow_rays: (shadow rays, those whose NEE term is zero whatever the walk finds, those of them that are occluded,
        all occluded ones) over n_samples samples of every stride-th pixel."""
        out = np.zeros(4, np.uint64)
        rng = np.ascontiguousarray(rng)
        self.lib.oracle_dead_shadow_rays(C.byref(config), C.byref(scene), _p(rng), C.c_uint32(n_samples), C.c_uint32(stride), _p(out))
        return tuple(int(v) for v in out)

    def bvh_build(self, vertices_xyzw, triangles, sah_samples=128):
        """BVHBuilder::new(vertices, indices).sah_samples(n).build() (reference: src/bvh.rs:59-324), restated in oracle/bvh_oracle.cpp.
        Returns (nodes, reordered triangles) as arrays of the input dtypes' layouts (nodes: 32-byte records as 8 x u32)."""
        v = np.ascontiguousarray(vertices_xyzw, np.float32).reshape(-1, 4)
        t = np.ascontiguousarray(triangles).copy()
        assert t.dtype.itemsize == 16
        nodes = np.zeros((2 * len(t) - 1, 8), np.uint32)
        n = C.c_size_t()
        rc = self.lib.oracle_bvh_build(_p(v), C.c_size_t(len(v)), _p(t), C.c_size_t(len(t)), C.c_uint32(sah_samples), _p(nodes),
                                       C.c_size_t(len(nodes)), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"oracle_bvh_build failed: {rc}")
        return nodes[:n.value].copy(), t

    def light_table(self, vertices_xyzw, triangles, materials):
        """build_light_pick_table (reference: src/light_pick.rs:13-122), restated in oracle/bvh_oracle.cpp -> (n, 28-byte records as 7 x u32)."""
        v = np.ascontiguousarray(vertices_xyzw, np.float32).reshape(-1, 4)
        t = np.ascontiguousarray(triangles)
        m = np.ascontiguousarray(materials)
        assert t.dtype.itemsize == 16 and m.dtype.itemsize == 96
        out = np.zeros((max(1, len(t)), 7), np.uint32)
        self.lib.oracle_light_table.restype = C.c_long
        n = self.lib.oracle_light_table(_p(v), C.c_size_t(len(v)), _p(t), C.c_size_t(len(t)), _p(m), C.c_size_t(len(m)), _p(out), C.c_size_t(len(out)))
        if n < 0:
            raise RuntimeError("oracle_light_table failed")
        return out[:n].copy()

    def trace_rays(self, scene, mode, origins, dirs, max_t=None):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        max_t = np.zeros(n, np.float32) if max_t is None else np.ascontiguousarray(max_t, np.float32)
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.uint32)
        flags = np.zeros(n, np.uint32)
        err = self.lib.oracle_trace_rays(C.byref(scene), C.c_int(mode), C.c_size_t(n), _p(origins), _p(dirs), _p(max_t),
                                         _p(t), _p(tri), _p(flags))
        return t, tri, flags, err

    def resolve(self, accum, sample_count, op):
        """mean + display tonemap (reference: src/trace.rs:303-308, src/resources/render.wgsl:36-153)."""
        accum = np.ascontiguousarray(accum, np.float32)
        out = np.zeros(accum.shape[:-1] + (3,), np.float32)
        self.lib.oracle_resolve(_p(accum), C.c_size_t(accum.size // 4), C.c_float(sample_count), C.c_uint32(op), _p(out))
        return out

    def lds(self, n, dim, offset):
        prod = C.c_uint32()
        v = self.lib.oracle_lds(C.c_uint32(n), C.c_uint32(dim), C.c_uint32(offset), C.byref(prod))
        return prod.value, float(v)

    def math(self, op, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = x if y is None else np.ascontiguousarray(y, np.float32)
        out = np.empty_like(x)
        rc = self.lib.oracle_math(C.c_int(op), _p(x), _p(y), _p(out), C.c_size_t(x.size))
        assert rc == 0
        return out

    def bsdf(self, kind, items):
        """Lambertian / Glass (kernels/src/bsdf.rs:46-176) restated: (n, 16) float32 in -> (n, 8) out (see oracle_bsdf)."""
        items = np.ascontiguousarray(items, np.float32).reshape(-1, 16)
        out = np.zeros((len(items), 8), np.float32)
        rc = self.lib.oracle_bsdf(C.c_int(kind), C.c_size_t(len(items)), _p(items), _p(out))
        assert rc == 0
        return out

    def sky(self, sun_direction4, origin3, dirs):
        sun = np.ascontiguousarray(sun_direction4, np.float32)
        org = np.ascontiguousarray(origin3, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros_like(dirs)
        self.lib.oracle_sky(_p(sun), _p(org), _p(dirs), _p(out), C.c_size_t(len(dirs)))
        return out
