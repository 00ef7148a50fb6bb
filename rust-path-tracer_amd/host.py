"""Python face of librpt_host.so — the host side of the render dispatch.

Names follow the reference's host interface for this path:
  World.from_path          <-> World::from_path      (reference: src/asset.rs:55-224)
  blue_noise_seeds         <-> rng_data_blue          (reference: src/trace.rs:150-157)
  TracingConfig defaults   <-> TracingConfig::default (reference: shared_structs/src/lib.rs:27-42)
  setup_trace / trace_gpu  <-> setup_trace / trace_gpu (reference: src/trace.rs:331-344, 136-224)
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import (BVH_NODE_DTYPE, LIGHT_PICK_DTYPE, MATERIAL_DTYPE, PER_VERTEX_DTYPE, RNG_DTYPE, TRIANGLE_DTYPE,
                   TracingConfig, WorldView, ptr)

_lib = None


def lib():
    """Load librpt_host.so (built in-tree by `make host` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        path = os.environ.get("RPT_HOST_LIB") or os.path.join(_ffi.LIB_DIR, "librpt_host.so")   # (RPT_HOST_LIB: a sanitizer build, tools/fuzz_glb.py)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.rpt_host_last_error.restype = C.c_char_p
        L.rpt_tracing_state_new.restype = C.c_void_p
        L.rpt_setup_trace.restype = C.c_void_p
        L.rpt_tracing_state_config.restype = C.POINTER(TracingConfig)
        L.rpt_tracing_state_config.argtypes = [C.c_void_p]
        L.rpt_tracing_state_framebuffer.restype = C.POINTER(C.c_float)
        L.rpt_tracing_state_framebuffer.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.rpt_tracing_state_samples.restype = C.c_uint32
        L.rpt_tracing_state_samples.argtypes = [C.c_void_p]
        L.rpt_tracing_state_free.argtypes = [C.c_void_p]
        L.rpt_tracing_state_set_sync_rate.argtypes = [C.c_void_p, C.c_uint32]
        L.rpt_tracing_state_set_running.argtypes = [C.c_void_p, C.c_int]
        L.rpt_tracing_state_set_dirty.argtypes = [C.c_void_p, C.c_int]
        L.rpt_tracing_state_set_overlap.argtypes = [C.c_void_p, C.c_int]
        L.rpt_tracing_state_set_interacting.argtypes = [C.c_void_p, C.c_int]
        L.rpt_tracing_state_copy_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rpt_tracing_state_set_config.argtypes = [C.c_void_p, C.POINTER(TracingConfig)]
        L.rpt_tracing_state_new.argtypes = [C.c_uint32, C.c_uint32]
        L.rpt_trace_gpu.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_char_p]
        L.rpt_world_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.rpt_world_load_ex.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.rpt_skybox_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.rpt_host_free.argtypes = [C.c_void_p]
        L.rpt_host_free.restype = None
        L.rpt_world_view_get.argtypes = [C.c_void_p, C.POINTER(WorldView)]
        L.rpt_world_free.argtypes = [C.c_void_p]
        L.rpt_world_save.argtypes = [C.c_void_p, C.c_char_p]
        L.rpt_world_load_cache.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.rpt_world_from_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.POINTER(C.c_void_p)]
        L.rpt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise HostError(f"librpt_host error {rc}: {lib().rpt_host_last_error().decode()}")


def default_config(width=1280, height=720, **overrides):
    """TracingConfig::default() with width/height set (TracingState::make_view_dependent_state)."""
    c = TracingConfig()
    lib().rpt_tracing_config_default(C.byref(c))
    c.width, c.height = width, height
    for k, v in overrides.items():
        if isinstance(v, (tuple, list)):
            for i, x in enumerate(v):
                getattr(c, k)[i] = x
        else:
            setattr(c, k, v)
    return c


def _view_array(addr, count, dtype):
    if not addr or count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class World:
    """The five POD buffers (+ optional atlas) the kernels consume (reference: src/asset.rs:9-16)."""

    def __init__(self, per_vertex, indices, nodes, materials, light_pick, atlas=None, bvh_max_depth=0,
                 n_emissive_triangles=0):
        self.per_vertex = np.ascontiguousarray(per_vertex, PER_VERTEX_DTYPE)
        self.indices = np.ascontiguousarray(indices, TRIANGLE_DTYPE)
        self.nodes = np.ascontiguousarray(nodes, BVH_NODE_DTYPE)
        self.materials = np.ascontiguousarray(materials, MATERIAL_DTYPE)
        self.light_pick = np.ascontiguousarray(light_pick, LIGHT_PICK_DTYPE)
        self.atlas = atlas  # HxWx4 uint8 or None
        self.bvh_max_depth = bvh_max_depth
        self.n_emissive_triangles = n_emissive_triangles

    @classmethod
    def _from_handle(cls, handle):
        v = WorldView()
        _check(lib().rpt_world_view_get(handle, C.byref(v)))
        atlas = None
        if v.atlas_rgba8:
            atlas = _view_array(v.atlas_rgba8, v.atlas_w * v.atlas_h * 4, np.dtype("u1")).reshape(v.atlas_h, v.atlas_w, 4)
        w = cls(_view_array(v.per_vertex, v.n_vertices, PER_VERTEX_DTYPE),
                _view_array(v.indices, v.n_triangles, TRIANGLE_DTYPE),
                _view_array(v.nodes, v.n_nodes, BVH_NODE_DTYPE),
                _view_array(v.materials, v.n_materials, MATERIAL_DTYPE),
                _view_array(v.light_pick, v.n_light_pick, LIGHT_PICK_DTYPE),
                atlas, v.bvh_max_depth, v.n_emissive_triangles)
        lib().rpt_world_free(handle)
        return w

    @classmethod
    def from_path(cls, path, emissive_strength=False):
        """World::from_path (reference: src/asset.rs:55-224).  emissive_strength=True honours
        KHR_materials_emissive_strength instead of the reference's fixed x15 (opt-in: not the reference's behaviour)."""
        h = C.c_void_p()
        _check(lib().rpt_world_load_ex(os.fsencode(path), C.c_uint32(1 if emissive_strength else 0), C.byref(h)))
        return cls._from_handle(h)

    @classmethod
    def from_cache(cls, path):
        """Load a ".rptscene" file (the five POD buffers written by World.save / the Rust host)."""
        h = C.c_void_p()
        _check(lib().rpt_world_load_cache(os.fsencode(path), C.byref(h)))
        return cls._from_handle(h)

    def save(self, path):
        """Write this World's buffers as ".rptscene" (byte-identical exchange format, include/rpt/rpt_host.h)."""
        header = np.zeros(1, np.dtype([("magic", "S8"), ("n", "<u8", 5), ("aw", "<u4"), ("ah", "<u4")]))
        header["magic"] = b"RPTSCN01"
        header["n"] = [len(self.per_vertex), len(self.indices), len(self.nodes), len(self.materials), len(self.light_pick)]
        atlas = b""
        if self.atlas is not None:
            header["ah"], header["aw"] = self.atlas.shape[:2]
            atlas = np.ascontiguousarray(self.atlas, np.uint8).tobytes()
        with open(path, "wb") as f:
            f.write(header.tobytes())
            for a in (self.per_vertex, self.indices, self.nodes, self.materials, self.light_pick):
                f.write(a.tobytes())
            f.write(atlas)

    @classmethod
    def from_buffers(cls, vertices, normals, uvs, triangles, materials):
        """Procedural scenes: run the reference's BVH / light-table / packing steps on raw geometry."""
        vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        normals = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        uvs = None if uvs is None else np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        triangles = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 4)
        materials = np.ascontiguousarray(materials, MATERIAL_DTYPE)
        h = C.c_void_p()
        _check(lib().rpt_world_from_buffers(ptr(vertices), ptr(normals), ptr(uvs), C.c_size_t(len(vertices)),
                                            ptr(triangles), C.c_size_t(len(triangles)), ptr(materials),
                                            C.c_size_t(len(materials)), C.byref(h)))
        return cls._from_handle(h)


def fixture(name):
    return os.path.join(_ffi.FIXTURES, name)


def write_png(path, rgb, srgb=True):
    """8-bit PNG of a resolved (H, W, 3) float frame, sRGB-encoded like the reference's saved renders."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    _check(lib().rpt_write_png(os.fsencode(path), ptr(rgb), C.c_uint32(rgb.shape[1]), C.c_uint32(rgb.shape[0]), int(bool(srgb))))


def load_skybox(path):
    """load_dynamic_image + dynamic_image_to_cpu_buffer (reference: src/asset.rs:238-273): the skybox file as the CPU
    path sees it — 8 bits per channel, alpha 1 — as an (H, W, 4) float32 array for Renderer.upload_scene(skybox_f32=...)."""
    p = C.POINTER(C.c_float)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().rpt_skybox_load(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib().rpt_host_free(p)


def blue_noise_tile(png_path=None):
    out = np.zeros(256 * 256, np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().rpt_blue_noise_tile(os.fsencode(png_path or fixture("bluenoise.png")), ptr(out), C.c_size_t(out.size),
                                     C.byref(w), C.byref(h)))
    return out[: w.value * h.value].reshape(h.value, w.value)


def blue_noise_seeds(width, height, png_path=None):
    """rng[i] = (0, seed(x % 256, y % 256)) as a (H*W,) RNG_DTYPE array (reference: src/trace.rs:150-157)."""
    out = np.zeros(width * height, RNG_DTYPE)
    _check(lib().rpt_blue_noise_seeds(os.fsencode(png_path or fixture("bluenoise.png")), C.c_uint32(width),
                                      C.c_uint32(height), ptr(out)))
    return out


class TracingState:
    """TracingState handle (reference: src/trace.rs:40-92) created by setup_trace."""

    def __init__(self, handle):
        self._h = handle

    @property
    def config(self):
        return lib().rpt_tracing_state_config(self._h).contents

    @property
    def samples(self):
        return lib().rpt_tracing_state_samples(self._h)

    def framebuffer(self):
        n = C.c_size_t()
        lib().rpt_tracing_state_framebuffer(self._h, C.byref(n))
        cfg = self.config
        out = np.empty(n.value, np.float32)
        _check(lib().rpt_tracing_state_copy_framebuffer(self._h, ptr(out), n.value))      # locked copy: the render thread may be publishing
        return out.reshape(cfg.height, cfg.width, 3)

    def set_sync_rate(self, n):
        lib().rpt_tracing_state_set_sync_rate(self._h, C.c_uint32(n))

    def set_running(self, running):
        lib().rpt_tracing_state_set_running(self._h, C.c_int(1 if running else 0))

    def set_dirty(self, dirty=True):
        lib().rpt_tracing_state_set_dirty(self._h, C.c_int(1 if dirty else 0))

    def set_interacting(self, on=True):
        """state.interacting (src/trace.rs:50): while up, every batch flushes (camera drag)."""
        lib().rpt_tracing_state_set_interacting(self._h, C.c_int(1 if on else 0))

    def set_overlap(self, on=True):
        """trace_gpu reads batch k back while batch k+1 renders (rpt_tracing_state_set_overlap)."""
        lib().rpt_tracing_state_set_overlap(self._h, C.c_int(1 if on else 0))

    def set_config(self, config):
        """state.config.write() while trace_gpu runs on another thread (locked copy); follow with set_dirty()."""
        lib().rpt_tracing_state_set_config(self._h, C.byref(config))

    def close(self):
        if self._h:
            lib().rpt_tracing_state_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def setup_trace(width, height, samples):
    """setup_trace(width, height, samples) (reference: src/trace.rs:331-344), exact sample count."""
    return TracingState(C.c_void_p(lib().rpt_setup_trace(C.c_uint32(width), C.c_uint32(height), C.c_uint32(samples))))


def trace_gpu(scene_path, skybox_path, state, device_id=0):
    """trace_gpu(scene_path, skybox_path, state) (reference: src/trace.rs:136-224) on librpt_hip.so."""
    _check(lib().rpt_trace_gpu(os.fsencode(scene_path), None if skybox_path is None else os.fsencode(skybox_path),
                               state._h, device_id, None))


def bvh_build(vertices_xyzw, triangles, sah_samples=128):
    """BVHBuilder::build on the host (rpt_bvh_build; reference src/bvh.rs:59-324): returns (nodes, reordered triangles)."""
    v = np.ascontiguousarray(vertices_xyzw, np.float32).reshape(-1, 4)
    t = np.ascontiguousarray(triangles, TRIANGLE_DTYPE).copy()
    nodes = np.zeros(max(1, 2 * len(t) - 1), BVH_NODE_DTYPE)
    n_nodes = C.c_size_t(0)
    L = lib()
    L.rpt_bvh_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    _check(L.rpt_bvh_build(v.ctypes.data, len(v), t.ctypes.data, len(t), sah_samples, nodes.ctypes.data, len(nodes), C.byref(n_nodes)))
    return nodes[: n_nodes.value].copy(), t


def set_bvh_builder(use_gpu, hip_library_path=None, device=0):
    """Choose the BVH builder behind World.from_path / from_buffers: the host restatement (default) or the device build
    (rpt_bvh_build_gpu).  Same output either way."""
    L = lib()
    L.rpt_host_set_bvh_builder.argtypes = [C.c_int, C.c_char_p, C.c_int]
    _check(L.rpt_host_set_bvh_builder(1 if use_gpu else 0, os.fsencode(hip_library_path) if hip_library_path else None, device))
