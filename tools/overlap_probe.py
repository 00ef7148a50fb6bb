"""Does overlapping two half-size pipelines on one GPU hide the drain tails of small launches?
K renderers on ONE GPU, each owning 1/(N*K) of DarkCornell 1024^2 on its own stream (ranks 0..K-1 of N*K), against one
renderer owning 1/N.  usage: python tools/overlap_probe.py N K [K ...]"""
import importlib, sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
w = rpt.World.from_path(rpt.fixture('DarkCornell.glb'))
cfg = rpt.default_config(1024, 1024); seeds = rpt.blue_noise_seeds(1024, 1024)
N = int(sys.argv[1])
for K in [int(a) for a in sys.argv[2:]]:
    rs = []
    for k in range(K):
        r = hip.Renderer(0, rank=k, world_size=N * K); r.upload_scene(w); r.set_config(cfg); r.reset(seeds); rs.append(r)
    for r in rs: r.render(32)
    best = None
    for rep in range(3):
        s0 = [r.stats() for r in rs]; t = time.perf_counter()
        for _ in range(8):
            for r in rs: r.render_async(32)
        for r in rs: r.wait()
        dt = time.perf_counter() - t; s1 = [r.stats() for r in rs]
        rays = sum(b['extension_rays'] - a['extension_rays'] for a, b in zip(s0, s1))
        best = max(best or 0, rays / dt / 1e6)
    print(f'1/{N} of the image as {K} pipeline(s): {best:.0f} Mrays/s on the GPU')
    for r in rs: r.close()
