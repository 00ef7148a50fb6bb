#!/usr/bin/env python3
"""Startup (scene preparation) time, the reference's other benchmarked quantity (benches/benchmark.rs:11-16:
"Startup time (GPU)" = trace_gpu with 0 samples, ~3.0 s on BreakTime.glb, which is absent).  Times World construction
from raw buffers — BVH build + light table + packing — with the sequential host builder and with the device builder,
on the 1 M-triangle BreakTime stand-in and on PBRTest."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rpt = importlib.import_module("rust-path-tracer_amd")
hip = importlib.import_module("rust-path-tracer_amd.hip")
host = importlib.import_module("rust-path-tracer_amd.host")
from scenes import deep_bvh_scene, scatter_scene  # noqa: E402

for kind, n in (("clustered", 200_000), ("clustered", 1_000_000), ("scattered", 1_000_000)):
    # built once with the host builder just to get the soup; "scattered": small triangles, leaves of one or two -> 2 M nodes
    w = deep_bvh_scene(n) if kind == "clustered" else scatter_scene(n)
    v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
    t = w.indices[np.random.default_rng(3).permutation(len(w.indices))]
    t0 = time.perf_counter(); hn, ht = host.bvh_build(v, t); th = time.perf_counter() - t0
    hip.bvh_build_gpu(v[:3], np.zeros(1, t.dtype))   # context / module load outside the timed call
    t0 = time.perf_counter(); gn, gt, ms = hip.bvh_build_gpu(v, t); tg = time.perf_counter() - t0
    same = hn.tobytes() == gn.tobytes() and ht.tobytes() == gt.tobytes()
    print(f"BVH build, {kind} {len(t)} triangles -> {len(hn)} nodes: host {th*1e3:8.1f} ms | GPU {tg*1e3:7.1f} ms wall, {ms:7.1f} ms device | identical: {same}")
