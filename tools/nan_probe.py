"""NaN / infinite / degenerate inputs (GPU box): camera, rotation, sun, lobe clamp set to NaN, inf, zero or absurd values, a NaN
vertex and a NaN box — the HIP path must neither hang nor differ from the oracle (NaN pixels compared as NaN).
usage: python tools/nan_probe.py"""
import importlib, sys, os, copy
import numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
from oracle_ffi import Oracle
orc = Oracle()
base = rpt.World.from_path(rpt.fixture('DarkCornell.glb'))
veach = rpt.World.from_path(rpt.fixture('VeachMIS.glb'))
W,H,spp=64,48,3
seeds = rpt.blue_noise_seeds(W,H)
nan=float('nan'); inf=float('inf')
cases = {
 'cam nan': dict(cam_position=(nan,1.0,-5.0,0.0)),
 'cam inf': dict(cam_position=(0.0,inf,-5.0,0.0)),
 'rot nan': dict(cam_rotation=(nan,0.0,0.0,0.0)),
 'sun nan': dict(sun_direction=(nan,1.0,0.0,15.0)),
 'sun zero': dict(sun_direction=(0.0,0.0,0.0,15.0)),
 'sun inf w': dict(sun_direction=(0.5,1.3,1.0,inf)),
 'clamp nan': dict(specular_weight_clamp=(nan,nan)),
 'clamp inverted': dict(specular_weight_clamp=(0.9,0.1)),
 'huge cam': dict(cam_position=(1e30,1e30,-1e30,0.0)),
}
def same(a,b):
    a=np.asarray(a); b=np.asarray(b)
    na,nb=np.isnan(a),np.isnan(b)
    return np.array_equal(na,nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))
for wname,w in (('DarkCornell',base),('VeachMIS',veach)):
  for nee in (0,1):
    for name,over in cases.items():
        cfg = rpt.default_config(W,H,nee=nee,**over)
        r = hip.Renderer(0); r.upload_scene(w); r.set_config(cfg); r.reset(seeds); r.render(spp)
        acc,n = r.read_accum(); st=r.stats(); r.close()
        ref,_,so = orc.trace_cpu(cfg, orc.scene(w), seeds, spp)
        ok = same(acc,ref) and st['extension_rays']==so.extension_rays and st['shadow_rays']==so.shadow_rays
        print(f'{wname:12s} nee {nee} {name:16s}', 'ok' if ok else 'MISMATCH', 'nan px', int(np.isnan(ref[...,0]).sum()), so.extension_rays, st['extension_rays'])
# a NaN vertex / NaN box
for what in ('vertex','box'):
    w = copy.copy(base)
    for nm in ('per_vertex','indices','nodes','materials','light_pick'): setattr(w,nm,getattr(base,nm).copy())
    if what=='vertex': w.per_vertex['vertex'][10,0]=nan
    else: w.nodes['aabb_min'][3,1]=nan
    cfg = rpt.default_config(W,H,nee=1)
    r = hip.Renderer(0); r.upload_scene(w); r.set_config(cfg); r.reset(seeds); r.render(spp)
    acc,n = r.read_accum(); st=r.stats(); r.close()
    ref,_,so = orc.trace_cpu(cfg, orc.scene(w), seeds, spp)
    print('nan', what, 'ok' if same(acc,ref) and st['extension_rays']==so.extension_rays else 'MISMATCH', int(np.isnan(ref[...,0]).sum()))
