"""ctypes mirrors of include/rpt/shared_structs.h, rpt.h and rpt_host.h.

Layouts follow the reference's `shared_structs` crate
(reference: shared_structs/src/lib.rs:12-191); numpy dtypes of the same layouts
are provided so buffers can be inspected / generated from Python.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
FIXTURES = os.path.join(REPO_ROOT, "fixtures")


class TracingConfig(C.Structure):
    _fields_ = [
        ("cam_position", C.c_float * 4),
        ("cam_rotation", C.c_float * 4),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("min_bounces", C.c_uint32),
        ("max_bounces", C.c_uint32),
        ("sun_direction", C.c_float * 4),
        ("nee", C.c_uint32),
        ("has_skybox", C.c_uint32),
        ("specular_weight_clamp", C.c_float * 2),
    ]

    def copy(self):
        c = TracingConfig()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(TracingConfig))
        return c


assert C.sizeof(TracingConfig) == 80

MATERIAL_DTYPE = np.dtype([
    ("emissive", "<f4", 4), ("albedo", "<f4", 4), ("roughness", "<f4", 4), ("metallic", "<f4", 4),
    ("normals", "<f4", 4), ("has_albedo_texture", "<u4"), ("has_metallic_texture", "<u4"),
    ("has_roughness_texture", "<u4"), ("has_normal_texture", "<u4")])
PER_VERTEX_DTYPE = np.dtype([("vertex", "<f4", 4), ("normal", "<f4", 4), ("tangent", "<f4", 4),
                             ("uv0", "<f4", 2), ("uv1", "<f4", 2)])
LIGHT_PICK_DTYPE = np.dtype([("triangle_index_a", "<u4"), ("triangle_area_a", "<f4"), ("triangle_pick_pdf_a", "<f4"),
                             ("triangle_index_b", "<u4"), ("triangle_area_b", "<f4"), ("triangle_pick_pdf_b", "<f4"),
                             ("ratio", "<f4")])
BVH_NODE_DTYPE = np.dtype([("aabb_min", "<f4", 3), ("triangle_count", "<u4"), ("aabb_max", "<f4", 3),
                           ("left_or_first", "<u4")])
TRIANGLE_DTYPE = np.dtype([("v0", "<u4"), ("v1", "<u4"), ("v2", "<u4"), ("material", "<u4")])
RNG_DTYPE = np.dtype([("n", "<u4"), ("offset", "<u4")])
assert MATERIAL_DTYPE.itemsize == 96 and PER_VERTEX_DTYPE.itemsize == 64 and LIGHT_PICK_DTYPE.itemsize == 28
assert BVH_NODE_DTYPE.itemsize == 32 and TRIANGLE_DTYPE.itemsize == 16 and RNG_DTYPE.itemsize == 8


class WorldView(C.Structure):
    _fields_ = [
        ("per_vertex", C.c_void_p), ("n_vertices", C.c_size_t),
        ("indices", C.c_void_p), ("n_triangles", C.c_size_t),
        ("nodes", C.c_void_p), ("n_nodes", C.c_size_t),
        ("materials", C.c_void_p), ("n_materials", C.c_size_t),
        ("light_pick", C.c_void_p), ("n_light_pick", C.c_size_t),
        ("atlas_rgba8", C.c_void_p), ("atlas_w", C.c_uint32), ("atlas_h", C.c_uint32),
        ("bvh_max_depth", C.c_uint32), ("n_emissive_triangles", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("sky_evals", C.c_uint64), ("light_index_clamped", C.c_uint64), ("iterations", C.c_uint64),
        ("render_ms", C.c_double), ("kernel_ms", C.c_double * 8), ("kernel_launches", C.c_uint64 * 8),
        ("shadow_rays_elided", C.c_uint64),
    ]


STAGE_NAMES = ["generate", "traverse", "shade", "shadow", "sky", "complete"]


def ptr(arr):
    """void* of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return C.c_void_p(arr.ctypes.data)
