import importlib, os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
rpt = importlib.import_module("rust-path-tracer_amd"); hip = importlib.import_module("rust-path-tracer_amd.hip")
from scenes import deep_bvh_scene, scatter_scene
w = scatter_scene(1_000_000) if 'scatter' in sys.argv[1:] else deep_bvh_scene(1_000_000)
v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
t = w.indices[np.random.default_rng(3).permutation(len(w.indices))]
hip.bvh_build_gpu(v[:48], t[:16] % 48 if False else np.zeros(1, t.dtype))      # (module load outside the build)
n, tt, ms = hip.bvh_build_gpu(v, t); print(len(n), ms)
