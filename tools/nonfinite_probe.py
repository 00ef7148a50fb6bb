"""Find the pixels of a full-size render whose accumulators are not finite, and what the CPU oracle has there.
usage: python tools/nonfinite_probe.py [scene W H nee spp]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
rpt = importlib.import_module("rust-path-tracer_amd")
hip = importlib.import_module("rust-path-tracer_amd.hip")
from oracle_ffi import Oracle

scene, W, H, nee, spp = (sys.argv[1:] + ["VeachMIS", "1920", "1080", "1", "1024"][len(sys.argv) - 1:])[:5]
W, H, nee, spp = int(W), int(H), int(nee), int(spp)
world = rpt.World.from_path(rpt.fixture(scene + ".glb"))
cfg = rpt.default_config(W, H, nee=nee)
seeds = rpt.blue_noise_seeds(W, H)
r = hip.Renderer(0)
r.upload_scene(world); r.set_config(cfg); r.reset(seeds)
first_bad = {}
for b in range(spp // 32):
    r.render(32)
    a, s = r.read_accum()
    bad = np.argwhere(~np.isfinite(a).all(axis=2))
    for (y, x) in bad:
        first_bad.setdefault((int(x), int(y)), (b, a[y, x].copy()))
print("non-finite pixels:", len(first_bad))
orc = Oracle("rpt_math")
osc = orc.scene(world)
for (x, y), (b, val) in sorted(first_bad.items())[:12]:
    lo = b * 32
    # the oracle up to the batch before, and sample by sample through the batch
    ref, rng, _ = orc.trace_cpu(cfg, osc, seeds, lo, rect=(x, y, x + 1, y + 1)) if lo else (np.zeros((H, W, 4), np.float32), seeds.copy(), None)
    when = None
    acc = ref
    rr = rng
    for k in range(32):
        prev = acc[y, x].copy()
        acc, rr, _ = orc.trace_cpu(cfg, osc, rr, 1, accum=acc, rect=(x, y, x + 1, y + 1))
        if when is None and not np.isfinite(acc[y, x]).all():
            when = (lo + k, prev, acc[y, x].copy())
    print(f"pixel ({x},{y}) seed {seeds[y * W + x]}: GPU non-finite first in batch {b}: {val}; oracle after the same batch: {acc[y, x]}; oracle first non-finite: {when}")
a, s = r.read_accum()
full_ref_bad = 0
r.close()
