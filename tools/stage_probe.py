import importlib, sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ["RPT_STAGE_TIMING"]="1"
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
w = rpt.World.from_path(rpt.fixture('DarkCornell.glb'))
cfg = rpt.default_config(1024, 1024); seeds = rpt.blue_noise_seeds(1024, 1024)
for world in (8, 1):
    r = hip.Renderer(0, rank=0, world_size=world); r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
    r.render(16)
    s0 = r.stats(); t = time.perf_counter()
    for _ in range(8): r.render(16)
    dt = time.perf_counter() - t; s1 = r.stats()
    km = {k: (s1['kernel_ms'][k]-s0['kernel_ms'][k])/8 for k in s1['kernel_ms']}
    print(world, f"ms/step {dt/8*1e3:.3f}", {k: round(v,3) for k,v in km.items()}, "sum", round(sum(km.values()),3), "launches/step", (s1['kernel_launches']['traverse']-s0['kernel_launches']['traverse'])/8)
    r.close()
