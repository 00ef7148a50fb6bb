"""Hostile texture coordinates (GPU box): NaN, infinite, huge, negative and exactly-one uvs on the textured scene, with and
without the image skybox — atlas addressing must follow the oracle's (image_polyfill.rs:38-55 casts) without reading outside the
atlas.  python tools/uv_probe.py"""
import importlib, os, sys, copy
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
from oracle_ffi import Oracle
from scenes import textured_scene
orc = Oracle()
base, sky = textured_scene()
W, H, spp = 96, 64, 3
seeds = rpt.blue_noise_seeds(W, H)
uvname = [n for n in base.per_vertex.dtype.names if 'uv' in n][0]
poisons = {
 'nan': lambda u, r: u.__setitem__(r.random(u.shape) < 0.2, np.nan),
 'inf': lambda u, r: u.__setitem__(r.random(u.shape) < 0.2, np.inf),
 '-inf': lambda u, r: u.__setitem__(r.random(u.shape) < 0.2, -np.inf),
 'huge': lambda u, r: u.__imul__(np.float32(3e9)),
 '1e30': lambda u, r: u.__imul__(np.float32(1e30)),
 'negative': lambda u, r: u.__imul__(np.float32(-7.25)),
 'exactly 0/1': lambda u, r: u.__setitem__(slice(None), np.round(u)),
 'tiny': lambda u, r: u.__imul__(np.float32(1e-40)),
}
bad = 0
for name, p in poisons.items():
    w = copy.copy(base)
    for nm in ('per_vertex', 'indices', 'nodes', 'materials', 'light_pick'):
        setattr(w, nm, getattr(base, nm).copy())
    rng = np.random.default_rng(2)
    with np.errstate(all='ignore'):
        p(w.per_vertex[uvname], rng)
    for nee, has_sky in ((0, 1), (1, 0)):
        cfg = rpt.default_config(W, H, nee=nee, has_skybox=has_sky, cam_position=(0.0, 1.6, -4.0, 0.0), cam_rotation=(0.05, 0.1, 0.0, 0.0))
        r = hip.Renderer(0); r.upload_scene(w, skybox_f32=sky); r.set_config(cfg); r.reset(seeds); r.render(spp)
        acc, n = r.read_accum(); st = r.stats(); r.close()
        ref, _, so = orc.trace_cpu(cfg, orc.scene(w, skybox_f32=sky), seeds, spp)
        na, nb = np.isnan(acc), np.isnan(ref)
        ok = np.array_equal(na, nb) and np.array_equal(acc[~na].view(np.uint32), ref[~nb].view(np.uint32)) and st['extension_rays'] == so.extension_rays and st['shadow_rays'] == so.shadow_rays
        bad += 0 if ok else 1
        print(f"uv {name:12s} nee {nee} skybox {has_sky}: {'ok' if ok else 'MISMATCH'}")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
