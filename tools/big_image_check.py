"""BASELINE config[4] resolution (4096 x 4096, nee = MIS) on DarkCornell: 16.8 M pixels in one launch, windows compared
with the oracle bit for bit (the missing BreakTime scene is replaced: this checks the SIZE, tests/ check the stand-in)."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'oracle'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
from oracle_ffi import Oracle
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096      # e.g. 16384: 268 M pixels, slot and byte offsets past 2^32
w = rpt.World.from_path(rpt.fixture('DarkCornell.glb'))
cfg = rpt.default_config(W, H, nee=1); seeds = rpt.blue_noise_seeds(W, H)
r = hip.Renderer(0); r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
t = time.perf_counter(); r.render(4); a, s = r.read_accum(); dt = time.perf_counter() - t
st = r.stats()
print(f"{W}^2 x 4 spp nee=MIS:", f"{dt:.2f} s", "rays", st["extension_rays"] + st["shadow_rays"], "all pixels sampled:", bool(np.all(a[..., 3] == 4)))
orc = Oracle(); osc = orc.scene(w)
for rect in ((W // 2 - 48, H // 2 + 52, W // 2 - 8, H // 2 + 84), (W - 36, H - 26, W, H), (0, 0, 24, 16)):
    ref, _, _ = orc.trace_cpu(cfg, osc, seeds, 4, rect=rect)
    x0, y0, x1, y1 = rect
    print(rect, "bitwise equal:", np.array_equal(a[y0:y1, x0:x1].view(np.uint32), ref[y0:y1, x0:x1].view(np.uint32)))
