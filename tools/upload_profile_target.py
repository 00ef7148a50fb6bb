"""Target of `rocprofv3 --kernel-trace --stats`: rpt_upload_scene of the 1 M-triangle clustered stand-in, three times (which kernels is the upload made of?)."""
import importlib, os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
from scenes import deep_bvh_scene, scatter_scene
w = deep_bvh_scene(1_000_000) if (len(sys.argv) < 2 or sys.argv[1] == 'deepbvh') else scatter_scene(1_000_000)
r = hip.Renderer(0)
for k in range(3):
    t = time.perf_counter(); r.upload_scene(w); dt = time.perf_counter() - t
    print(f'upload {k}: {dt * 1e3:.1f} ms, probe {r.shadow_order()["probe_ms"]:.2f} ms')
r.close()
