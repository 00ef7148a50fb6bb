#!/usr/bin/env python3
"""How long does build_light_pick_table (src/light_pick.rs:24-122 -> csrc/host/light_table.cpp) take when EVERY triangle of a 1 M-triangle
scene is emissive — the worst case for the one scene-preparation step that still runs on the host (SURVEY.md 8f N1's second
half)?  Compared with the BVH build of the same scene (host and device figures: profiles/r02_startup_bench.txt).  CPU only."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rpt = importlib.import_module("rust-path-tracer_amd")
host = importlib.import_module("rust-path-tracer_amd.host")
ffi = importlib.import_module("rust-path-tracer_amd._ffi")
from scenes import scatter_scene  # noqa: E402

L = host.lib()
L.rpt_light_table_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
for n in (100_000, 1_000_000):
    w = scatter_scene(n)
    v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
    mats = w.materials.copy()
    mats["emissive"][:, :3] = np.random.default_rng(1).uniform(0.5, 20.0, (len(mats), 3)).astype(np.float32)      # every material emits
    table = np.zeros(len(w.indices), ffi.LIGHT_PICK_DTYPE)
    n_out = C.c_size_t()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = L.rpt_light_table_build(v.ctypes.data, len(v), w.indices.ctypes.data, len(w.indices), mats.ctypes.data, len(mats), table.ctypes.data, len(table), C.byref(n_out))
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    t0 = time.perf_counter(); host.bvh_build(v, w.indices); tb = time.perf_counter() - t0
    print(f"{len(w.indices)} triangles, all emissive: light table of {n_out.value} entries in {best * 1e3:.1f} ms on the host (one thread); "
          f"host BVH build of the same triangles {tb * 1e3:.0f} ms")
