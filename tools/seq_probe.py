"""DarkCornell 1024^2, 32-spp batches with chosen bounce limits (for tools/seq_probe.sh: per-launch durations of one batch).
usage: python tools/seq_probe.py MIN_BOUNCES MAX_BOUNCES [NEE]"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rpt = importlib.import_module("rust-path-tracer_amd"); hip = importlib.import_module("rust-path-tracer_amd.hip")
mn, mx = int(sys.argv[1]), int(sys.argv[2]); nee = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w = rpt.World.from_path(rpt.fixture("DarkCornell.glb"))
cfg = rpt.default_config(1024, 1024, min_bounces=mn, max_bounces=mx, nee=nee); seeds = rpt.blue_noise_seeds(1024, 1024)
r = hip.Renderer(0); r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
for _ in range(6):
    r.render_async(32)
r.wait(); r.close()
