"""Device BVH build of a 1 M-triangle stand-in with its triangles shuffled and in the leaf order of an earlier build (what a scene file gives).
usage: python tools/bvh_order_probe.py [scatter]"""
import importlib, os, sys, time
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
rpt = importlib.import_module("rust-path-tracer_amd"); hip = importlib.import_module("rust-path-tracer_amd.hip")
from scenes import deep_bvh_scene, scatter_scene
w = scatter_scene(1_000_000) if 'scatter' in sys.argv[1:] else deep_bvh_scene(1_000_000)
v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
ts = w.indices[np.random.default_rng(3).permutation(len(w.indices))]
hip.bvh_build_gpu(v[:48], np.zeros(1, ts.dtype))
for name, t in (("shuffled", ts), ("bvh order", w.indices), ("shuffled", ts), ("bvh order", w.indices)):
    t0 = time.perf_counter(); n, tt, ms = hip.bvh_build_gpu(v, t); print(name, len(n), "device ms", round(ms, 2), "wall", round((time.perf_counter() - t0) * 1e3, 2))
