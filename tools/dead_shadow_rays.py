#!/usr/bin/env python3
"""How many shadow rays decide nothing?  (CPU analysis.)  sample_direct_lighting (kernels/src/light_pick.rs:100-173) traces its shadow ray BEFORE it
looks at light_pdf and bsdf_pdf (:141-158): when the picked light point faces away from the surface point (light_pdf = 0) or lies below the surface's
horizon (bsdf_pdf = 0) the term is zero whether the ray is occluded or not — `radiance += mask_nan(throughput * 0)` leaves every bit of radiance as it
was (radiance is never -0.0: it starts at +0.0 and x + y is -0.0 only for two negative zeros).
usage: python tools/dead_shadow_rays.py [scene ...]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_ffi import Oracle, _p  # noqa: E402

rpt = importlib.import_module("rust-path-tracer_amd")
SCENES = {"DarkCornell": (1024, 1024, {"nee": 1}), "VeachMIS": (1920, 1080, {"nee": 1}), "FurnaceTest": (256, 256, {"nee": 1}),
          "deepbvh": (2048, 2048, {"nee": 1, "cam_position": (0.0, 2.5, -0.5, 0.0)}), "scatter": (2048, 2048, {"nee": 1, "cam_position": (0.0, 1.8, -0.9, 0.0)})}
orc = Oracle()
for name in sys.argv[1:] or ["DarkCornell", "VeachMIS", "FurnaceTest"]:
    W, H, over = SCENES[name]
    if name in ("deepbvh", "scatter"):
        from scenes import deep_bvh_scene, scatter_scene
        world = (deep_bvh_scene if name == "deepbvh" else scatter_scene)(1_000_000)
    else:
        world = rpt.World.from_path(rpt.fixture(name + ".glb"))
    cfg = rpt.default_config(W, H, **over)
    out = np.zeros(4, np.uint64)
    stride = 8 if W > 300 else 2
    sc, seeds = orc.scene(world), rpt.blue_noise_seeds(W, H)
    orc.lib.oracle_dead_shadow_rays(C.byref(cfg), C.byref(sc), _p(seeds), C.c_uint32(4), C.c_uint32(stride), _p(out))
    n, dead, dead_occ, occ = (int(v) for v in out)
    print(f"{name:12s} {n:9d} shadow rays: {100 * dead / n:5.1f} % add a zero term whatever the walk finds ({100 * dead_occ / max(dead, 1):.0f} % of those are occluded); "
          f"occluded {100 * occ / n:.1f} %, occluded among the ones that matter {100 * (occ - dead_occ) / max(n - dead, 1):.1f} %")
