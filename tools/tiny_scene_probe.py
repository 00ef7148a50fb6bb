"""Degenerate scene shapes (GPU box): one triangle (the root is a leaf), two, three; no emissive triangle at all with every
NEE mode; everything emissive — image, rng and ray counts against the oracle.  python tools/tiny_scene_probe.py"""
import importlib, os, sys
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
from oracle_ffi import Oracle
orc = Oracle()

def world(n_tris, emissive):
    rng = np.random.default_rng(n_tris)
    c = np.stack([rng.uniform(-1, 1, n_tris), rng.uniform(0.5, 1.5, n_tris), rng.uniform(-1, 1, n_tris)], 1)
    a = rng.normal(size=(n_tris, 3)) * 0.6; b = rng.normal(size=(n_tris, 3)) * 0.6
    verts = np.stack([c - a, c + a, c + b], 1).reshape(-1, 3).astype(np.float32)
    p = verts.reshape(-1, 3, 3)
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]); fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    normals = np.repeat(fn, 3, axis=0).astype(np.float32)
    tris = np.concatenate([np.arange(3 * n_tris).reshape(-1, 3), (np.arange(n_tris) % 2).reshape(-1, 1)], 1).astype(np.uint32)
    m = np.zeros(2, rpt._ffi.MATERIAL_DTYPE)
    m["albedo"][:] = [[0.8, 0.4, 0.3, 1], [0.3, 0.6, 0.8, 1]]
    m["roughness"][:, :] = 0.5
    if emissive == "all": m["emissive"][:] = [[5, 4, 3, 15], [2, 3, 5, 15]]
    elif emissive == "some": m["emissive"][1] = [5, 4, 3, 15]
    return rpt.World.from_buffers(verts, normals, None, tris, m)

bad = 0
W, H, spp = 72, 56, 3
seeds = rpt.blue_noise_seeds(W, H)
for n_tris in (1, 2, 3, 5):
    for emissive in ("none", "some", "all"):
        w = world(n_tris, emissive)
        for nee in (0, 1, 2):
            cfg = rpt.default_config(W, H, nee=nee, cam_position=(0.0, 1.0, -4.0, 0.0))
            r = hip.Renderer(0); r.upload_scene(w); r.set_config(cfg); r.reset(seeds); r.render(spp)
            acc, n = r.read_accum(); st = r.stats(); rn = r.read_rng(); r.close()
            ref, rng_ref, so = orc.trace_cpu(cfg, orc.scene(w), seeds, spp)
            ok = (np.array_equal(acc.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rn["n"], rng_ref["n"]) and
                  (st["extension_rays"], st["shadow_rays"], st["sky_evals"]) == (so.extension_rays, so.shadow_rays, so.sky_evals))
            bad += 0 if ok else 1
            print(f"{n_tris} triangle(s), {len(w.nodes)} node(s), emissive {emissive:4s} nee {nee}: {'ok' if ok else 'MISMATCH'}")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
