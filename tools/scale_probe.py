"""Strong-scaling probe on ONE GPU: rank 0 of N renders its 1/N of DarkCornell 1024^2 (the other ranks' tiles are simply
not rendered), which gives the per-GPU rate a real N-GPU run would see before the gather.  Two batch sizes per N: the
reference's sync_rate of 32 samples (src/trace.rs:75) and 32 x N — the batch bench.py runs at N GPUs since round 6: with up
to 256 samples of a pixel in flight a rank's launches cover as many slots as the whole image's do at 32.
usage: python tools/scale_probe.py [N ...]   (default 1 2 4 8)"""
import importlib, sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip')
w = rpt.World.from_path(rpt.fixture('DarkCornell.glb'))
cfg = rpt.default_config(1024, 1024); seeds = rpt.blue_noise_seeds(1024, 1024)
ref = None
for world in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    for spp in sorted({32, 32 * world}):
        r = hip.Renderer(0, rank=0, world_size=world); r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
        r.render(spp); r.render(spp)
        steps = max(2, 256 // spp) if world > 1 else 8
        best = None
        for rep in range(3):
            s0 = r.stats(); t = time.perf_counter()
            for _ in range(steps): r.render_async(spp)
            r.wait()
            dt = time.perf_counter() - t; s1 = r.stats()
            rays = s1['extension_rays'] - s0['extension_rays']
            kms = {k: round((s1['kernel_ms'][k] - s0['kernel_ms'][k]) / steps, 3) for k in s1['kernel_ms']}   # with RPT_STAGE_TIMING=1
            if best is None or dt < best[0]: best = (dt, rays, kms)
        dt, rays, kms = best
        rate = rays / dt / 1e6
        if world == 1 and spp == 32: ref = rate
        print(f'world {world}, {spp:3d} spp per batch: {rate:.0f} Mrays/s per GPU' + (f' = {100 * rate / ref:.1f} % of the whole image' if ref else '') +
              f' ; ms/step {dt/steps*1e3:.3f} ; stage ms/step {kms}')
        r.close()
