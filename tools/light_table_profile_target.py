#!/usr/bin/env python3
"""rocprofv3 target: rpt_light_table_build_gpu on the 1 M-triangle scattered stand-in with every triangle emissive (kernel trace of the device passes)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
hip = importlib.import_module("rust-path-tracer_amd.hip")
from scenes import scatter_scene  # noqa: E402

w = scatter_scene(1_000_000)
v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
mats = w.materials.copy()
mats["emissive"][:, :3] = np.random.default_rng(1).uniform(0.5, 20.0, (len(mats), 3)).astype(np.float32)
for _ in range(3):
    table, n_em, ms = hip.light_table_build_gpu(v, w.indices, mats)
    print(len(table), n_em, {k: round(x, 2) for k, x in ms.items()})
