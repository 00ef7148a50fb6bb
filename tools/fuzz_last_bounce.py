#!/usr/bin/env python3
"""Randomised differential test (GPU box) aimed at the hit-or-miss walks of a batch's last extension rays (k_traverse.h k_traverse_nearest_stream LAST):
no NEE, scenes that live in LDS (DarkCornell, with random further materials made emissive — 2 .. many emissive triangles, NaN and negative emission
included — and the open textured scene with and without its image skybox), random size / camera / bounce limits / sample counts on both sides of the
known-length limit / samples in flight / sun, and the order forced to each of its five settings or left to the probe.  The HIP path must equal the
oracle bit for bit.  python tools/fuzz_last_bounce.py [N] [seed]"""
import copy
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
rpt = importlib.import_module("rust-path-tracer_amd")
hip = importlib.import_module("rust-path-tracer_amd.hip")
from oracle_ffi import Oracle  # noqa: E402
from scenes import textured_scene  # noqa: E402

MODES = [None, ("RPT_LAST_ORDER", "off"), ("RPT_LAST_ORDER", "near"), ("RPT_LAST_ORDER", "opaque"), ("RPT_LAST_ORDER", "small"), ("RPT_LAST_ORDER", "ratio")]


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
    orc = Oracle()
    cornell = rpt.World.from_path(rpt.fixture("DarkCornell.glb"))
    textured, sky_img = textured_scene()
    bad = 0
    seen_modes = {}
    for case in range(n_cases):
        for var in ("RPT_LAST_ORDER",):
            os.environ.pop(var, None)
        mode = MODES[rng.integers(len(MODES))]
        if mode:
            os.environ[mode[0]] = mode[1]
        if rng.integers(0, 4) == 0:
            name, w, sky = "textured", textured, sky_img
        else:
            name, sky = "cornell", None
            w = copy.copy(cornell)
            w.materials = cornell.materials.copy()
            for m in range(len(w.materials)):
                k = rng.integers(0, 10)
                if k == 0:
                    w.materials["emissive"][m] = (float(rng.uniform(0, 20)), float(rng.uniform(0, 20)), float(rng.uniform(0, 20)), 0.0)
                elif k == 1:
                    w.materials["emissive"][m] = (0.0, 0.0, 0.0, 0.0)
                elif k == 2 and rng.integers(0, 4) == 0:
                    w.materials["emissive"][m] = (float("nan") if rng.integers(0, 2) else -3.0, 0.0, 1.0, 0.0)
        W, H = int(rng.integers(1, 220)), int(rng.integers(1, 160))
        max_b = int(rng.integers(1, 7))
        min_b = int(rng.integers(0, 7))
        spp = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 33, 40]))
        s_in_flight = int(rng.choice([0, 0, 1, 2, 4, 8, 16, 32]))
        cam = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0.3, 2.5)), float(rng.uniform(-6, -1)), 0.0)
        rot = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.8, 0.8)), 0.0, 0.0)
        has_sky = int(sky is not None and rng.integers(0, 2))
        over = {}
        if rng.integers(0, 2):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            over["sun_direction"] = (float(d[0]), float(d[1]), float(d[2]), float(rng.choice([0.0, 1.0, 15.0, 40.0])))
        if name == "cornell" and rng.integers(0, 3):
            cam = (0.0, 1.0, float(rng.uniform(-4.5, -2.0)), 0.0)          # inside / in front of the box: most paths stay in it
            rot = (float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.2, 0.2)), 0.0, 0.0)
        cfg = rpt.default_config(W, H, nee=0, min_bounces=min_b, max_bounces=max_b, cam_position=cam, cam_rotation=rot, has_skybox=has_sky, **over)
        seeds = rpt.blue_noise_seeds(W, H)
        r = hip.Renderer(0)
        r.set_samples_in_flight(s_in_flight)
        r.upload_scene(w, skybox_f32=sky if has_sky else None)
        lo = r.last_bounce_order()
        seen_modes[lo["mode"]] = seen_modes.get(lo["mode"], 0) + 1
        r.set_config(cfg); r.reset(seeds)
        first = int(rng.integers(0, spp + 1))
        r.render(first); r.render(spp - first)
        acc, n = r.read_accum(); g = r.stats(); r.close()
        ref, _, st = orc.trace_cpu(cfg, orc.scene(w, skybox_f32=sky if has_sky else None), seeds, spp)
        na, nb = np.isnan(acc), np.isnan(ref)
        ok = (n == spp and np.array_equal(na, nb) and np.array_equal(acc[~na].view(np.uint32), ref[~nb].view(np.uint32)) and g["extension_rays"] == st.extension_rays
              and g["sky_evals"] == st.sky_evals)
        print(f"{case:3d} {name:9s} {W}x{H} spp {first}+{spp - first} bounces {min_b}/{max_b} S {s_in_flight} sky {has_sky} emissive triangles {lo['emissive_triangles']} "
              f"mode {lo['mode']} ({mode[1] if mode else 'probe'}): {'ok' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
    print("modes seen (0 whole walk, 1 near first, 2-4 fixed rules):", dict(sorted(seen_modes.items())))
    print("mismatches:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
