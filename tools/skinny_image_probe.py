"""Images at the limits of the 16-bit pixel coordinates (GPU box): 65535 x 1, 1 x 65535, 70 x 65535 ... as one rank and as three
(the partial images add up to the oracle's); 65536 and 0 are refused.  python tools/skinny_image_probe.py"""
import importlib, sys, os
import numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
from oracle_ffi import Oracle
orc=Oracle(); w=rpt.World.from_path(rpt.fixture('DarkCornell.glb')); osc=orc.scene(w)
for W,H in ((65535,1),(1,65535),(65535,2),(3,40000),(70,65535)):
    for world_size in (1,3):
        cfg=rpt.default_config(W,H,nee=1); seeds=rpt.blue_noise_seeds(W,H)
        ref,_,so=orc.trace_cpu(cfg,osc,seeds,2,threads=8)
        full=np.zeros((H,W,4),np.float32)
        for rank in range(world_size):
            r=hip.Renderer(0,rank=rank,world_size=world_size); r.upload_scene(w); r.set_config(cfg); r.reset(seeds); r.render(2)
            a,n=r.read_accum(); r.close(); full+=a
        print(W,H,'ranks',world_size,'ok' if np.array_equal(full.view(np.uint32),ref.view(np.uint32)) else 'MISMATCH')
        bad = globals().get('bad', 0) + (0 if np.array_equal(full.view(np.uint32),ref.view(np.uint32)) else 1)
for W,H in ((65536,4),(4,65536),(0,5)):
    r=hip.Renderer(0); r.upload_scene(w)
    try:
        r.set_config(rpt.default_config(W,H)); print(W,H,'ACCEPTED?!'); bad = globals().get('bad', 0) + 1
    except hip.RptError as e:
        print(W,H,'rejected:',str(e)[:60])
    r.close()
print('mismatches:', globals().get('bad', 0))
sys.exit(1 if globals().get('bad', 0) else 0)
