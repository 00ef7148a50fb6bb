#!/usr/bin/env python3
"""Randomised API-SEQUENCE test (GPU box): one long-lived context is driven through random, valid sequences of the C ABI —
render / render_async of 0..7 samples, read_accum / map_accum / read_rng, new configuration, reset with fresh seeds or as a
resume (accum = mean x samples), a different scene, samples-in-flight changes, the local communicator's gather / read_gathered,
wait — while a model advances the CPU oracle by the same samples.  Whenever the image is read it must equal the model's,
bit for bit, and so must the sample count and the ray counters.   python tools/fuzz_api.py [steps] [seed]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
rpt = importlib.import_module("rust-path-tracer_amd")
hip = importlib.import_module("rust-path-tracer_amd.hip")
from oracle_ffi import Oracle  # noqa: E402


def main(steps=None, seed=None, quiet=False):
    steps = steps if steps is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 300)
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 7))
    orc = Oracle()
    worlds = {n: rpt.World.from_path(rpt.fixture(n + ".glb")) for n in ("DarkCornell", "VeachMIS", "FurnaceTest")}
    names = sorted(worlds)
    r = hip.Renderer(0)
    local_comm = bool(rng.integers(0, 2))
    if local_comm:
        r.comm_init_local()
    m = {}                                                      # the model: what the context must hold

    def new_config():
        W, H = int(rng.integers(1, 120)), int(rng.integers(1, 90))
        nee = int(rng.integers(0, 3))
        cam = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0.4, 2.5)), float(rng.uniform(-6, -1)), 0.0)
        m["cfg"] = rpt.default_config(W, H, nee=nee, max_bounces=int(rng.integers(1, 5 if nee == 0 else 4)), min_bounces=int(rng.integers(0, 4)),
                                      cam_position=cam, cam_rotation=(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.6, 0.6)), 0.0, 0.0))
        r.set_config(m["cfg"])

    def new_scene():
        m["scene"] = names[rng.integers(len(names))]
        r.upload_scene(worlds[m["scene"]])
        m["osc"] = orc.scene(worlds[m["scene"]])

    def reset():
        cfg = m["cfg"]
        W, H = cfg.width, cfg.height
        seeds = rpt.blue_noise_seeds(W, H)
        if rng.integers(0, 3) == 0:                             # rng_data_uniform-style seeds (src/trace.rs:158)
            seeds = seeds.copy()
            seeds["n"] = rng.integers(0, 2 ** 32, W * H, dtype=np.uint64).astype(np.uint32)
            seeds["offset"] = 0
        if rng.integers(0, 4) == 0:                             # resume: accum = mean x samples (src/trace.rs:163-164)
            k = int(rng.integers(1, 9))
            init = (rng.random((H, W, 4)).astype(np.float32) * np.float32(k))
            init[..., 3] = k
            r.reset(seeds, accum_init=init, samples_init=k)
            m["accum"], m["samples"] = init.copy(), k
        else:
            r.reset(seeds)
            m["accum"], m["samples"] = np.zeros((H, W, 4), np.float32), 0
        m["rng"] = np.ascontiguousarray(seeds).copy()
        m["rays"] = [0, 0, 0]
        m["stats0"] = r.stats()
        m["snap"] = None

    def render(n, asynchronous):
        (r.render_async if asynchronous else r.render)(n)
        if n:
            m["accum"], m["rng"], st = orc.trace_cpu(m["cfg"], m["osc"], m["rng"], n, accum=m["accum"])
            assert st.error_flags == 0
            m["samples"] += n
            for i, v in enumerate((st.extension_rays, st.shadow_rays, st.sky_evals)):
                m["rays"][i] += v

    def check(img, samples, what):
        ok = samples == m["samples"] and np.array_equal(np.asarray(img).view(np.uint32), m["accum"].view(np.uint32))
        if not ok:
            raise AssertionError(f"{what}: image / sample count differ from the model ({samples} vs {m['samples']})")

    new_scene(); new_config(); reset()
    log = []
    for step in range(steps):
        op = rng.choice(["render", "render", "render_async", "render_async", "read", "map", "rng", "stats", "config", "reset", "scene",
                         "in_flight", "gather", "read_gathered", "wait"])
        log.append(op)
        if op in ("render", "render_async"):
            render(int(rng.integers(0, 8)), op == "render_async")
        elif op == "read":
            check(*r.read_accum(), "read_accum")
        elif op == "map":
            check(*r.map_accum(), "map_accum")
        elif op == "rng":
            assert np.array_equal(r.read_rng()["n"], m["rng"]["n"]), "rng"
        elif op == "stats":
            r.wait()
            s, s0 = r.stats(), m["stats0"]
            got = [s["extension_rays"] - s0["extension_rays"], s["shadow_rays"] - s0["shadow_rays"], s["sky_evals"] - s0["sky_evals"]]
            assert got == m["rays"], ("ray counters", got, m["rays"])
        elif op == "config":
            new_config(); reset()
        elif op == "reset":
            reset()
        elif op == "scene":
            new_scene(); reset()
        elif op == "in_flight":
            r.set_samples_in_flight(int(rng.choice([0, 1, 2, 4, 8, 32, 64, 256])))
            reset()
        elif op == "gather" and local_comm:
            r.gather_async()
            m["snap"] = (m["accum"].copy(), m["samples"])
        elif op == "read_gathered" and local_comm and m["snap"] is not None:
            img, s = r.read_gathered()
            assert s == m["snap"][1] and np.array_equal(img.view(np.uint32), m["snap"][0].view(np.uint32)), "read_gathered"
        elif op == "wait":
            r.wait()
    check(*r.read_accum(), "final read_accum")
    r.close()
    if not quiet:
        print(f"{steps} API calls ({'with' if local_comm else 'without'} the local communicator): consistent with the model")
    return 0


if __name__ == "__main__":
    sys.exit(main())
