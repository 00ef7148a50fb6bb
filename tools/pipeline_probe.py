"""What do K un-fenced pipelines per rank (rpt_comm_add_pipeline) buy where a rank's launches are small?  One GPU, a local
communicator (no RCCL), an image with as many pixels as 1/N of DarkCornell 1024^2, 32-spp batches, ONE gather per batch through the
library (snapshot per pipeline, un-tile on the second stream) — the loop bench.py runs per rank, minus the exchange.
usage: python tools/pipeline_probe.py [N ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rpt = importlib.import_module("rust-path-tracer_amd"); hip = importlib.import_module("rust-path-tracer_amd.hip")
w = rpt.World.from_path(rpt.fixture("DarkCornell.glb"))
for N in [int(a) for a in sys.argv[1:]] or [8, 4, 1]:
    W, H = 1024, 1024 // N
    cfg = rpt.default_config(W, H); seeds = rpt.blue_noise_seeds(W, H)
    for K in (1, 2, 3):
        owner = hip.Renderer(0); owner.comm_init_local()
        ctxs = [owner] + [hip.Renderer(0) for _ in range(K - 1)]
        for e in ctxs[1:]:
            owner.comm_add_pipeline(e)
        for p in ctxs:
            p.upload_scene(w); p.set_config(cfg); p.reset(seeds); p.render(32); p.reset(seeds)
        best = 0.0
        for rep in range(4):
            s0 = [p.stats() for p in ctxs]; t = time.perf_counter()
            for _ in range(16):
                for p in ctxs:
                    p.render_async(32)
                owner.gather_async()
            owner.gather_wait()
            for p in ctxs:
                p.wait()
            dt = time.perf_counter() - t; s1 = [p.stats() for p in ctxs]
            rays = sum(b["extension_rays"] - a["extension_rays"] for a, b in zip(s0, s1))
            best = max(best, rays / dt / 1e6)
        print(f"{W}x{H} (1/{N} of 1024^2), {K} pipeline(s), gather per batch: {best:.0f} Mrays/s")
        for p in reversed(ctxs):
            p.close()
