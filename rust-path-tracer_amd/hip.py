"""Renderer: ctypes over the rpt.h C ABI of librpt_hip.so (the MI355X wavefront path tracer).

Every method maps 1:1 to a C entry point, which in turn cites the reference
call it replaces (include/rpt/rpt.h; reference: src/trace.rs:136-224).
There is deliberately NO fallback: if the HIP library is missing or no GPU is
present this module raises.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import RNG_DTYPE, Stats, TracingConfig, ptr

_lib = None

EXPORTS = [
    "rpt_abi_version", "rpt_create", "rpt_set_partition", "rpt_set_samples_in_flight", "rpt_upload_scene", "rpt_set_config", "rpt_reset",
    "rpt_render", "rpt_render_async", "rpt_wait", "rpt_stream", "rpt_read_accum", "rpt_resolve", "rpt_read_rng", "rpt_local_pixels", "rpt_local_block_device_ptr",
    "rpt_rank_pixels", "rpt_tile_order", "rpt_untile", "rpt_get_stats", "rpt_destroy", "rpt_last_error",
    "rpt_comm_init_local", "rpt_debug_math", "rpt_debug_math_host", "rpt_debug_math_sweep", "rpt_debug_trace_rays", "rpt_debug_bsdf", "rpt_bvh_build_gpu", "rpt_light_table_build_gpu",
    "rpt_map_accum", "rpt_comm_unique_id", "rpt_comm_init", "rpt_comm_world", "rpt_gather_async", "rpt_gather_wait", "rpt_read_gathered",
    "rpt_gathered_device_ptr", "rpt_multi_create", "rpt_multi_size", "rpt_multi_ctx", "rpt_multi_upload_scene", "rpt_multi_set_config",
    "rpt_multi_reset", "rpt_multi_render", "rpt_multi_wait", "rpt_multi_read_accum", "rpt_multi_get_stats", "rpt_multi_destroy",
    "rpt_multi_last_error", "rpt_comm_library", "rpt_debug_trace_rays_production", "rpt_build_fingerprint", "rpt_debug_comm_selftest", "rpt_device_info", "rpt_shadow_order", "rpt_debug_shadow_order_host", "rpt_last_bounce_order", "rpt_debug_last_order_host", "rpt_debug_short_batch",
]
COMM_ID_BYTES = 128
MULTI_ALLOW_SHARED_DEVICE = 1


def lib_path():
    # RPT_HIP_LIB: developer override used for A/B runs of differently tuned builds
    return os.environ.get("RPT_HIP_LIB") or os.path.join(_ffi.LIB_DIR, "librpt_hip.so")


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make hip` (or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(path)
        L.rpt_last_error.restype = C.c_char_p
        L.rpt_last_error.argtypes = [C.c_void_p]
        L.rpt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rpt_destroy.argtypes = [C.c_void_p]
        L.rpt_destroy.restype = None
        L.rpt_set_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.rpt_set_samples_in_flight.argtypes = [C.c_void_p, C.c_int]
        L.rpt_upload_scene.argtypes = [C.c_void_p] + [C.c_void_p, C.c_size_t] * 5 + [C.c_void_p, C.c_uint32, C.c_uint32] * 2
        L.rpt_set_config.argtypes = [C.c_void_p, C.POINTER(TracingConfig)]
        L.rpt_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.rpt_render.argtypes = [C.c_void_p, C.c_uint32]
        L.rpt_render_async.argtypes = [C.c_void_p, C.c_uint32]
        L.rpt_wait.argtypes = [C.c_void_p]
        L.rpt_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rpt_read_accum.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.rpt_read_rng.argtypes = [C.c_void_p, C.c_void_p]
        L.rpt_resolve.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.rpt_local_pixels.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.rpt_local_block_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rpt_rank_pixels.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        L.rpt_untile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.rpt_tile_order.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
        L.rpt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rpt_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.rpt_debug_math_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.rpt_debug_math_sweep.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.rpt_debug_trace_rays.argtypes = [C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 6
        L.rpt_debug_trace_rays_production.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5
        L.rpt_bvh_build_gpu.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
        L.rpt_debug_bsdf.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rpt_map_accum.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32)]
        L.rpt_comm_unique_id.argtypes = [C.c_void_p]
        L.rpt_build_fingerprint.restype = C.c_char_p
        L.rpt_build_fingerprint.argtypes = []
        L.rpt_comm_library.restype = C.c_char_p
        L.rpt_comm_library.argtypes = []
        L.rpt_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rpt_comm_init_local.argtypes = [C.c_void_p]
        L.rpt_device_info.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.rpt_shadow_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.rpt_last_bounce_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.rpt_debug_comm_selftest.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        L.rpt_comm_world.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.rpt_gather_async.argtypes = [C.c_void_p]
        L.rpt_gather_wait.argtypes = [C.c_void_p]
        L.rpt_read_gathered.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.rpt_gathered_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rpt_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        L.rpt_multi_size.argtypes = [C.c_void_p]
        L.rpt_multi_ctx.argtypes = [C.c_void_p, C.c_int]
        L.rpt_multi_ctx.restype = C.c_void_p
        L.rpt_multi_upload_scene.argtypes = [C.c_void_p] + [C.c_void_p, C.c_size_t] * 5 + [C.c_void_p, C.c_uint32, C.c_uint32] * 2
        L.rpt_multi_set_config.argtypes = [C.c_void_p, C.POINTER(TracingConfig)]
        L.rpt_multi_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.rpt_multi_render.argtypes = [C.c_void_p, C.c_uint32]
        L.rpt_multi_wait.argtypes = [C.c_void_p]
        L.rpt_multi_read_accum.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.rpt_multi_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rpt_multi_destroy.argtypes = [C.c_void_p]
        L.rpt_multi_destroy.restype = None
        L.rpt_multi_last_error.argtypes = [C.c_void_p]
        L.rpt_multi_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class RptError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"librpt_hip error {code}: {message}")
        self.code = code


class Renderer:
    """One rpt_ctx on one GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device_id=0, rank=0, world_size=1):
        self._h = C.c_void_p()
        rc = lib().rpt_create(device_id, C.byref(self._h))
        if rc != 0:
            raise RptError(rc, lib().rpt_last_error(None).decode())
        self.rank, self.world_size = rank, world_size
        if world_size != 1:
            self._check(lib().rpt_set_partition(self._h, rank, world_size))
        self.config = None

    @classmethod
    def borrowed(cls, handle, config=None, rank=0, world_size=1):
        """A view of an rpt_ctx somebody else owns (rpt_multi_ctx): every method works, close() / garbage collection leave the context alone."""
        self = cls.__new__(cls)
        self._h, self._borrowed = handle, True
        self.rank, self.world_size, self.config = rank, world_size, config
        return self

    def _check(self, rc):
        if rc != 0:
            raise RptError(rc, lib().rpt_last_error(self._h).decode())

    def set_partition(self, rank, world_size):
        """rpt_set_partition: this context renders the tiles of `rank` of `world_size` (state is re-allocated at the next reset)."""
        self._check(lib().rpt_set_partition(self._h, rank, world_size))
        self.rank, self.world_size = rank, world_size

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                lib().rpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_samples_in_flight(self, samples):
        """0 = automatic. Never changes the result (tests/test_gpu_parity.py::test_samples_in_flight_invisible)."""
        self._check(lib().rpt_set_samples_in_flight(self._h, samples))

    # -- rpt_upload_scene <-> World::into_gpu (reference: src/asset.rs:226-235)
    def upload_scene(self, world, skybox_f32=None):
        atlas = getattr(world, "atlas", None)
        aw = ah = sw = sh = 0
        if atlas is not None:
            atlas = np.ascontiguousarray(atlas, np.uint8)
            ah, aw = atlas.shape[:2]
        if skybox_f32 is not None:
            skybox_f32 = np.ascontiguousarray(skybox_f32, np.float32)
            sh, sw = skybox_f32.shape[:2]
        self._check(lib().rpt_upload_scene(
            self._h, ptr(world.per_vertex), len(world.per_vertex), ptr(world.indices), len(world.indices),
            ptr(world.nodes), len(world.nodes), ptr(world.materials), len(world.materials),
            ptr(world.light_pick), len(world.light_pick), ptr(atlas), aw, ah, ptr(skybox_f32), sw, sh))

    # -- rpt_set_config <-> config_buffer write (reference: src/trace.rs:168,219)
    def set_config(self, config):
        self._check(lib().rpt_set_config(self._h, C.byref(config)))
        self.config = config.copy()

    # -- rpt_reset <-> rng/output buffer creation + flush (reference: src/trace.rs:164-170,219-221)
    def reset(self, rng_seed, accum_init=None, samples_init=0):
        rng_seed = np.ascontiguousarray(rng_seed, RNG_DTYPE)
        assert rng_seed.size == self.config.width * self.config.height
        if accum_init is not None:
            accum_init = np.ascontiguousarray(accum_init, np.float32)
            assert accum_init.size == rng_seed.size * 4
        self._check(lib().rpt_reset(self._h, ptr(rng_seed), ptr(accum_init), samples_init))

    # -- rpt_render <-> the enqueue/poll loop (reference: src/trace.rs:182-194)
    def render(self, n_samples):
        self._check(lib().rpt_render(self._h, n_samples))

    def render_async(self, n_samples):
        """rpt_render_async: returns once the batch is enqueued when its iteration count is known (n_samples <= slots per
        pixel), else behaves like render()."""
        self._check(lib().rpt_render_async(self._h, n_samples))

    def wait(self):
        self._check(lib().rpt_wait(self._h))

    def stream_ptr(self):
        """The hipStream_t the library enqueues on (wrap with torch.cuda.ExternalStream to order work after a batch)."""
        p = C.c_void_p()
        self._check(lib().rpt_stream(self._h, C.byref(p)))
        return p.value

    # -- rpt_read_accum <-> output_buffer.read_blocking (reference: src/trace.rs:198)
    def read_accum(self, out=None):
        if out is None:
            out = np.zeros((self.config.height, self.config.width, 4), np.float32)
        assert out.dtype == np.float32 and out.size == self.config.height * self.config.width * 4 and out.flags["C_CONTIGUOUS"]
        samples = C.c_uint32()
        self._check(lib().rpt_read_accum(self._h, ptr(out), C.byref(samples)))
        return out, samples.value

    def map_accum(self):
        """rpt_map_accum: the library's own pinned read-back buffer as an (H, W, 4) array — no copy; valid until the
        next read_accum / map_accum / set_config on this renderer."""
        p = C.POINTER(C.c_float)()
        samples = C.c_uint32()
        self._check(lib().rpt_map_accum(self._h, C.byref(p), C.byref(samples)))
        h, w = self.config.height, self.config.width
        return np.ctypeslib.as_array(p, shape=(h, w, 4)), samples.value

    def resolve(self, tonemap_op=0):
        """mean radiance (+ display tonemap 0..6, reference: src/resources/render.wgsl:131-153) as (H, W, 3) float32."""
        out = np.zeros((self.config.height, self.config.width, 3), np.float32)
        self._check(lib().rpt_resolve(self._h, tonemap_op, ptr(out)))
        return out

    def read_rng(self):
        out = np.zeros(self.config.height * self.config.width, RNG_DTYPE)
        self._check(lib().rpt_read_rng(self._h, ptr(out)))
        return out

    def stats(self):
        s = Stats()
        self._check(lib().rpt_get_stats(self._h, C.byref(s)))
        d = {k: getattr(s, k) for k in ("samples", "extension_rays", "shadow_rays", "shadow_rays_elided", "sky_evals", "light_index_clamped",
                                         "iterations", "render_ms")}
        d["shadow_rays_traced"] = d["shadow_rays"] - d["shadow_rays_elided"]       # walked on the device; shadow_rays counts as the reference does
        d["kernel_ms"] = {n: s.kernel_ms[i] for i, n in enumerate(_ffi.STAGE_NAMES)}
        d["kernel_launches"] = {n: s.kernel_launches[i] for i, n in enumerate(_ffi.STAGE_NAMES)}
        return d

    # -- multi-GPU gather support
    def local_pixels(self):
        n = C.c_uint64()
        self._check(lib().rpt_local_pixels(self._h, C.byref(n)))
        return n.value

    def rank_pixels(self, rank):
        n = C.c_uint64()
        self._check(lib().rpt_rank_pixels(self._h, rank, C.byref(n)))
        return n.value

    def local_block_device_ptr(self):
        p = C.c_void_p()
        self._check(lib().rpt_local_block_device_ptr(self._h, C.byref(p)))
        return p.value

    def untile(self, dev_blocks_ptr, dev_out_ptr, block_stride_pixels=0):
        self._check(lib().rpt_untile(self._h, C.c_void_p(dev_blocks_ptr), block_stride_pixels, C.c_void_p(dev_out_ptr)))

    # -- the gather inside the library (RCCL): rpt_comm_init / rpt_gather_async / rpt_read_gathered
    def comm_init(self, unique_id, rank, world_size):
        """ncclCommInitRank on this renderer's device + rpt_set_partition(rank, world_size); unique_id = the 128 bytes
        rank 0 got from comm_unique_id(), handed to every rank by any means."""
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(lib().rpt_comm_init(self._h, buf, rank, world_size))
        self.rank, self.world_size = rank, world_size

    def comm_init_local(self):
        """rpt_comm_init_local: a one-rank communicator without RCCL (overlapped read-back on one GPU)."""
        self._check(lib().rpt_comm_init_local(self._h))

    def shadow_order(self):
        """rpt_shadow_order: {"fixed": bool, "visits_near", "visits_fixed", "probe_rays"} — which (bit-exact) order the shadow walks of the scene use."""
        f, n = C.c_uint32(), C.c_uint32()
        vn, vf, ms = C.c_double(), C.c_double(), C.c_double()
        self._check(lib().rpt_shadow_order(self._h, C.byref(f), C.byref(vn), C.byref(vf), C.byref(n), C.byref(ms)))
        return {"fixed": bool(f.value), "visits_near": vn.value, "visits_fixed": vf.value, "probe_rays": n.value, "probe_ms": ms.value}

    LAST_BOUNCE_MODES = ("whole walk", "hit or miss, near child first", "hit or miss, more opaque child first", "hit or miss, smaller subtree first",
                         "hit or miss, more opaque per node first")

    def last_bounce_order(self):
        """rpt_last_bounce_order: how the last extension rays of a batch without NEE are walked on the uploaded scene (every mode gives the same image)."""
        m, ne, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
        v = (C.c_double * 4)()
        ms = C.c_double()
        self._check(lib().rpt_last_bounce_order(self._h, C.byref(m), C.byref(ne), v, C.byref(n), C.byref(ms)))
        return {"mode": m.value, "mode_is": self.LAST_BOUNCE_MODES[m.value], "emissive_triangles": ne.value, "probe_rays": n.value, "probe_ms": ms.value,
                "probe_node_visits": {"near child first": v[0], "more opaque first": v[1], "smaller subtree first": v[2], "more opaque per node first": v[3]}}

    def comm_world(self):
        r, w = C.c_uint32(), C.c_uint32()
        self._check(lib().rpt_comm_world(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_selftest(self, n_floats=1 << 20):
        """rpt_debug_comm_selftest: words that differ after a grouped ncclSend / ncclRecv ring (one rank: to and from itself)."""
        bad = C.c_uint64()
        self._check(lib().rpt_debug_comm_selftest(self._h, n_floats, C.byref(bad)))
        return bad.value

    def gather_async(self):
        self._check(lib().rpt_gather_async(self._h))

    def gather_wait(self):
        self._check(lib().rpt_gather_wait(self._h))

    def read_gathered(self, out=None):
        if out is None:
            out = np.zeros((self.config.height, self.config.width, 4), np.float32)
        samples = C.c_uint32()
        self._check(lib().rpt_read_gathered(self._h, ptr(out), C.byref(samples)))
        return out, samples.value

    # -- test hooks (include/rpt/rpt_debug.h)
    def debug_short_batch(self, on=True):
        """rpt_debug_short_batch: asynchronous batches enqueue one iteration too few (the completion checks must notice)."""
        lib().rpt_debug_short_batch.argtypes = [C.c_void_p, C.c_int]
        self._check(lib().rpt_debug_short_batch(self._h, 1 if on else 0))

    def debug_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = x if y is None else np.ascontiguousarray(y, np.float32)
        out = np.empty_like(x)
        self._check(lib().rpt_debug_math(self._h, op, ptr(x), ptr(y), ptr(out), x.size))
        return out

    def debug_math_sweep(self, op, lo_bits, count, y=1.0):
        """rpt_debug_math_sweep: (mismatches, first bad bit pattern) of a cheap exact operation vs its IEEE form."""
        bad, first = C.c_uint64(), C.c_uint32()
        self._check(lib().rpt_debug_math_sweep(self._h, op, lo_bits, count, y, C.byref(bad), C.byref(first)))
        return bad.value, first.value

    def debug_bsdf(self, kind, items):
        """Lambertian / Glass of kernels/src/bsdf.rs:46-176 on the device (rpt_debug_bsdf): (n, 16) in -> (n, 8) out."""
        items = np.ascontiguousarray(items, np.float32).reshape(-1, 16)
        out = np.zeros((len(items), 8), np.float32)
        self._check(lib().rpt_debug_bsdf(self._h, kind, len(items), ptr(items), ptr(out)))
        return out

    def debug_trace_rays(self, any_hit, origins, dirs, max_t=None):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        max_t = np.zeros(n, np.float32) if max_t is None else np.ascontiguousarray(max_t, np.float32)
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.uint32)
        flags = np.zeros(n, np.uint32)
        self._check(lib().rpt_debug_trace_rays(self._h, int(bool(any_hit)), n, ptr(origins), ptr(dirs), ptr(max_t),
                                               ptr(t), ptr(tri), ptr(flags)))
        return t, tri, flags

    def debug_trace_rays_production(self, origins, dirs):
        """rpt_debug_trace_rays_production: nearest hits through the traversal stage rpt_render itself launches for this scene / state."""
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.uint32)
        flags = np.zeros(n, np.uint32)
        self._check(lib().rpt_debug_trace_rays_production(self._h, n, ptr(origins), ptr(dirs), ptr(t), ptr(tri), ptr(flags)))
        return t, tri, flags


def comm_unique_id():
    """ncclGetUniqueId through the C ABI (rank 0 calls it; the 128 bytes go to every rank)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib().rpt_comm_unique_id(buf)
    if rc != 0:
        raise RptError(rc, lib().rpt_last_error(None).decode())
    return bytes(buf)


def device_info(device_id=0):
    """rpt_device_info: (compute units, peak engine clock in MHz) of a HIP device."""
    cus, khz = C.c_uint32(), C.c_uint32()
    rc = lib().rpt_device_info(device_id, C.byref(cus), C.byref(khz))
    if rc:
        raise RptError(rc, lib().rpt_last_error(None).decode())
    return cus.value, khz.value / 1e3


def build_fingerprint():
    """rpt_build_fingerprint: tools/source_fingerprint.py of the sources the loaded library was built from."""
    return lib().rpt_build_fingerprint().decode()


def comm_library():
    """rpt_comm_library: the collective library the process resolved ("" before the first communicator)."""
    return lib().rpt_comm_library().decode()


class MultiRenderer:
    """rpt_multi_*: ONE process driving several GPUs (ncclCommInitAll) — the entry points a single render thread like
    the reference's (src/trace.rs:136-224) calls; the caller sees one W x H image."""

    def __init__(self, device_ids, allow_shared_device=False):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        rc = lib().rpt_multi_create(ids, len(device_ids), MULTI_ALLOW_SHARED_DEVICE if allow_shared_device else 0, C.byref(self._h))
        if rc != 0:
            raise RptError(rc, lib().rpt_multi_last_error(None).decode())
        self.config = None

    def _check(self, rc):
        if rc != 0:
            raise RptError(rc, lib().rpt_multi_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().rpt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return lib().rpt_multi_size(self._h)

    def rank_view(self, rank):
        """The rpt_ctx of one rank as a (borrowed) Renderer: per-GPU statistics, stage times, partition queries."""
        return Renderer.borrowed(self.ctx_handle(rank), self.config, rank, self.size())

    def ctx_handle(self, rank):
        """rpt_multi_ctx: the borrowed rpt_ctx of one rank (for rpt_set_samples_in_flight / rpt_get_stats per GPU)."""
        lib().rpt_multi_ctx.restype = C.c_void_p
        lib().rpt_multi_ctx.argtypes = [C.c_void_p, C.c_int]
        return C.c_void_p(lib().rpt_multi_ctx(self._h, rank))

    def upload_scene(self, world, skybox_f32=None):
        atlas = getattr(world, "atlas", None)
        aw = ah = sw = sh = 0
        if atlas is not None:
            atlas = np.ascontiguousarray(atlas, np.uint8)
            ah, aw = atlas.shape[:2]
        if skybox_f32 is not None:
            skybox_f32 = np.ascontiguousarray(skybox_f32, np.float32)
            sh, sw = skybox_f32.shape[:2]
        self._check(lib().rpt_multi_upload_scene(
            self._h, ptr(world.per_vertex), len(world.per_vertex), ptr(world.indices), len(world.indices),
            ptr(world.nodes), len(world.nodes), ptr(world.materials), len(world.materials),
            ptr(world.light_pick), len(world.light_pick), ptr(atlas), aw, ah, ptr(skybox_f32), sw, sh))

    def set_config(self, config):
        self._check(lib().rpt_multi_set_config(self._h, C.byref(config)))
        self.config = config.copy()

    def reset(self, rng_seed, accum_init=None, samples_init=0):
        rng_seed = np.ascontiguousarray(rng_seed, RNG_DTYPE)
        if accum_init is not None:
            accum_init = np.ascontiguousarray(accum_init, np.float32)
        self._check(lib().rpt_multi_reset(self._h, ptr(rng_seed), ptr(accum_init), samples_init))

    def render(self, n_samples):
        self._check(lib().rpt_multi_render(self._h, n_samples))

    def wait(self):
        self._check(lib().rpt_multi_wait(self._h))

    def read_accum(self, out=None):
        if out is None:
            out = np.zeros((self.config.height, self.config.width, 4), np.float32)
        assert out.dtype == np.float32 and out.size == self.config.height * self.config.width * 4 and out.flags["C_CONTIGUOUS"]
        samples = C.c_uint32()
        self._check(lib().rpt_multi_read_accum(self._h, ptr(out), C.byref(samples)))
        return out, samples.value

    def stats(self):
        s = Stats()
        self._check(lib().rpt_multi_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k in ("samples", "extension_rays", "shadow_rays", "shadow_rays_elided", "sky_evals", "light_index_clamped", "iterations")}


def tile_order(width, height, rank, world_size):
    """(x | y << 16) of every pixel of `rank`'s tile-major block, in block order (no GPU needed)."""
    n = C.c_size_t()
    rc = lib().rpt_tile_order(width, height, rank, world_size, None, 0, C.byref(n))
    if rc != 0:
        raise RptError(rc, "rpt_tile_order")
    out = np.zeros(n.value, np.uint32)
    rc = lib().rpt_tile_order(width, height, rank, world_size, ptr(out), out.size, C.byref(n))
    if rc != 0:
        raise RptError(rc, "rpt_tile_order")
    return out


def last_order_host(world):
    """rpt_debug_last_order_host: the upload-time decision about the order of the hit-or-miss lanes of the last extension rays, without a GPU."""
    r, n = C.c_uint32(), C.c_uint32()
    v = (C.c_double * 4)()
    flip = np.zeros(max(1, (len(world.nodes) - 1) // 2), np.uint8)
    L = lib()
    L.rpt_debug_last_order_host.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_void_p]
    rc = L.rpt_debug_last_order_host(ptr(world.per_vertex), len(world.per_vertex), ptr(world.indices), len(world.indices), ptr(world.nodes), len(world.nodes),
                                     ptr(world.materials), len(world.materials), C.byref(r), v, C.byref(n), ptr(flip))
    if rc != 0:
        raise RptError(rc, "rpt_debug_last_order_host")
    return {"rule": r.value, "visits": [v[k] for k in range(4)], "probe_rays": n.value, "flip": flip}


def shadow_order_host(world):
    """rpt_debug_shadow_order_host: the upload-time decision about the shadow walks' order for a World, without a GPU."""
    f, n = C.c_uint32(), C.c_uint32()
    vn, vf = C.c_double(), C.c_double()
    flip = np.zeros(max(1, (len(world.nodes) - 1) // 2), np.uint8)
    L = lib()
    L.rpt_debug_shadow_order_host.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_void_p]
    rc = L.rpt_debug_shadow_order_host(ptr(world.per_vertex), len(world.per_vertex), ptr(world.indices), len(world.indices), ptr(world.nodes), len(world.nodes),
                                       ptr(world.materials), len(world.materials), ptr(world.light_pick), len(world.light_pick),
                                       C.byref(f), C.byref(vn), C.byref(vf), C.byref(n), ptr(flip))
    if rc != 0:
        raise RptError(rc, "rpt_debug_shadow_order_host")
    return {"fixed": bool(f.value), "visits_near": vn.value, "visits_fixed": vf.value, "probe_rays": n.value, "flip": flip}


def debug_math_host(op, x, y=None):
    """The host build of rpt_math.h inside librpt_hip.so (no GPU needed)."""
    x = np.ascontiguousarray(x, np.float32)
    y = x if y is None else np.ascontiguousarray(y, np.float32)
    out = np.empty_like(x)
    rc = lib().rpt_debug_math_host(op, ptr(x), ptr(y), ptr(out), x.size)
    if rc != 0:
        raise RptError(rc, "rpt_debug_math_host")
    return out


def light_table_build_gpu(vertices_xyzw, triangles, materials, device=0):
    """build_light_pick_table on the GPU (rpt_light_table_build_gpu; reference src/light_pick.rs:13-122).
    Returns (table as LIGHT_PICK_DTYPE array, number of emissive triangles, {"total", "device", "host_chains", "transfers"} in milliseconds)."""
    from ._ffi import LIGHT_PICK_DTYPE, MATERIAL_DTYPE, TRIANGLE_DTYPE
    v = np.ascontiguousarray(vertices_xyzw, np.float32).reshape(-1, 4)
    t = np.ascontiguousarray(triangles, TRIANGLE_DTYPE)
    m = np.ascontiguousarray(materials, MATERIAL_DTYPE)
    table = np.empty(max(1, len(t)), LIGHT_PICK_DTYPE)          # (np.zeros would fault in 28 bytes per triangle of pages the call overwrites)
    n, n_em = C.c_size_t(0), C.c_uint32(0)
    ms = (C.c_double * 4)()
    L = lib()
    L.rpt_light_table_build_gpu.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    rc = L.rpt_light_table_build_gpu(device, v.ctypes.data, len(v), t.ctypes.data, len(t), m.ctypes.data, len(m), table.ctypes.data, len(table),
                                     C.byref(n), C.byref(n_em), ms)
    if rc != 0:
        raise RptError(rc, L.rpt_last_error(None).decode())
    return table[: n.value], n_em.value, {"total": ms[0], "device": ms[1], "host_chains": ms[2], "transfers": ms[3]}


def bvh_build_gpu(vertices_xyzw, triangles, sah_samples=128, device=0):
    """BVHBuilder::build on the GPU (rpt_bvh_build_gpu; reference src/bvh.rs:59-324).
    vertices_xyzw: (n, 4) float32; triangles: TRIANGLE_DTYPE array (v0, v1, v2, material).
    Returns (nodes, reordered triangles, device milliseconds)."""
    from ._ffi import BVH_NODE_DTYPE, TRIANGLE_DTYPE
    v = np.ascontiguousarray(vertices_xyzw, np.float32).reshape(-1, 4)
    # the call reorders the triangles in place: a private copy, as words (numpy copies a structured array field by field: 6 ms for 16 MB instead of 1.5)
    t = np.ascontiguousarray(triangles, TRIANGLE_DTYPE).view(np.uint32).copy().view(TRIANGLE_DTYPE)
    nodes = np.empty(max(1, 2 * len(t) - 1), BVH_NODE_DTYPE)      # (written by the call; zeroing and copying 64 MB of node pool was 25 ms of a 1 M-triangle "startup" in this harness)
    n_nodes = C.c_size_t(0)
    ms = C.c_double(0.0)
    rc = lib().rpt_bvh_build_gpu(device, v.ctypes.data, len(v), t.ctypes.data, len(t), sah_samples, nodes.ctypes.data, len(nodes),
                                 C.byref(n_nodes), C.byref(ms))
    if rc != 0:
        raise RptError(rc, lib().rpt_last_error(None).decode())
    return nodes[: n_nodes.value], t, ms.value
