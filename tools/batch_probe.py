import importlib, sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
w = rpt.World.from_path(rpt.fixture('DarkCornell.glb'))
cfg = rpt.default_config(1024, 1024); seeds = rpt.blue_noise_seeds(1024, 1024)
spp = int(sys.argv[1]); S = int(sys.argv[2])
for world in (8, 1):
    r = hip.Renderer(0, rank=0, world_size=world)
    if S: r.set_samples_in_flight(S)
    r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
    r.render(spp)
    s0 = r.stats(); t = time.perf_counter()
    n = 256 // spp
    for _ in range(n): r.render_async(spp)
    r.wait()
    dt = time.perf_counter() - t; s1 = r.stats()
    rays = s1['extension_rays'] - s0['extension_rays']
    print(f'spp/batch {spp} S {S or "auto"} world {world}: {rays/dt/1e6:.0f} Mrays/s per GPU ; ms per 16 spp {dt/n*1e3*16/spp:.3f}')
    r.close()
