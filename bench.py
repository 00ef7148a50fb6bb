#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s of the MI355X wavefront path tracer.

Workload (BASELINE.json configs[1]): DarkCornell.glb, 1024x1024, 256 spp, default
TracingConfig (nee = 0, min/max bounces 3/4 — the reference's own bench setting,
benches/benchmark.rs:17-19), blue-noise seeds.  A "step" is one sample batch:
`rpt_render(spp_per_step)` over every pixel of this rank's tiles, followed (N > 1)
by the ONE gather of per-rank tile-major accumulator blocks to rank 0 and the
root's un-tile.  A batch is 32 samples — the reference's default `sync_rate` (src/trace.rs:75), the number of
samples its GPU loop renders between two read-backs; default 8 steps x 32 spp = the full 256 spp of the config.

  python bench.py --gpus 1 --steps 8 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = (extension + shadow rays traced by all
ranks in the K timed steps) / max-over-ranks wall time, scene and state already
resident in HBM.  Extra objects: `roofline` (dominant kernel, HIP-event timed in
the same run) and `cpu_baseline` (the CPU oracle — a port of the reference's
trace_cpu, test infrastructure — timed on this host's cores on a bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, total spp of the BASELINE config, config overrides)
    "darkcornell": ("DarkCornell", 1024, 1024, 256, {}),
    "darkcornell_mis": ("DarkCornell", 1024, 1024, 256, {"nee": 1}),
    "veachmis": ("VeachMIS", 1920, 1080, 1024, {"nee": 1}),
    "pbrtest": ("PBRTest", 2048, 2048, 512, {}),
    # BASELINE config 4 AS IT IS WRITTEN ("PBRTest.glb with albedo/normal/rough/metal textures 2048x2048"): the shipped file carries no
    # texture, so its own buffers + a labelled SYNTHETIC 4096^2 RGBA8 atlas laid out by the reference's packer (tests/scenes.py
    # pbrtest_textured_scene: four maps per material through the reference's own uvst fields)
    "pbrtest_textured": ("synthetic-textures:PBRTest", 2048, 2048, 512, {}),
    "furnace": ("FurnaceTest", 256, 256, 16, {}),
    # BASELINE config 5 names BreakTime.glb, absent from the reference mount: labelled procedural stand-in
    # (tests/scenes.py: 1 M clustered long thin triangles in a lit room; deep BVH, fat leaves)
    "deepbvh": ("procedural:deep_bvh_1M", 2048, 2048, 4096, {"nee": 1, "cam_position": (0.0, 2.5, -0.5, 0.0)}),
    # the other face of a large scene: 1 M SMALL scattered triangles -> 2 M nodes, leaves of one or two triangles
    # (32-bit stack entries in the walks; the clustered stand-in above has fat leaves and only 31 k nodes)
    "scatter": ("procedural:scatter_1M", 2048, 2048, 4096, {"nee": 1, "cam_position": (0.0, 1.8, -0.9, 0.0)}),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def algorithmic_bytes(n_ext, n_shadow, n_mis, n_samples):
    """minimal SoA wavefront traffic, every record written once and read once."""
    return 128 * n_ext + 96 * n_shadow + 128 * n_mis + 40 * n_samples


# bytes the traversal stage itself must move per extension ray: the two 16-byte ray records
# {origin, direction, stage word} read, the 8-byte hit record {t, triangle|backface} written
TRAVERSE_BYTES_PER_RAY = 32 + 8


def usable_cores():
    """Host threads this process may actually run: affinity mask, capped by the cgroup CPU quota (the GPU box
    shows 256 hardware threads but cpu.max = 16 CPUs; more threads than that only adds throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


# algorithmic bytes per unit of each stage's OWN traffic and the unit it is counted in
STAGE_MODEL = {
    "traverse": ("extension_rays", TRAVERSE_BYTES_PER_RAY),          # hit word 8 + ray 24 read, hit record 8 written
    "shade": ("extension_rays", 96),                                  # hit 8 + ray 24 + thr 16 read, ray 24 + hit 8 + thr 16 written
    "shadow": ("shadow_rays_traced", 80),                                    # entry 32 + contribution 16 read, radiance 16 + 16 RMW
    "sky": ("sky_evals", 76),                                         # slot id 4 + ray 24 + thr 16 + rad 16 read, rad 16 written
    "generate": ("samples", 80),
    "complete": ("samples", 72),                                      # per slot hit 8 + rad 16 read, hit 8 written; per pixel accum 32 + rng 16 (amortised)
}
VALU_ISSUE_CYCLES_FLOOR = 2.1   # SIMD cycles a wave64 v_fma_f32 occupies the issue port (tools/microbench/valu_rates.hip)


def data_label(scene):
    if scene.startswith("synthetic-textures:"):
        return ("fixtures/" + scene.split(":")[1] + ".glb (reference scene file: geometry, uvs, BVH, light table) + SYNTHETIC textures: the file has none "
                "(tests/scenes.py pbrtest_textured_scene: albedo / metallic / roughness / normal map per material in a 4096x4096 RGBA8 atlas)")
    if scene.startswith("procedural:"):
        return "synthetic stand-in for the missing BreakTime.glb (tests/scenes.py)"
    return "fixtures/" + scene + ".glb (reference scene file)"


def build_world(rpt, scene):
    if scene.startswith("synthetic-textures:"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from scenes import pbrtest_textured_scene
        return pbrtest_textured_scene()
    if scene.startswith("procedural:"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from scenes import deep_bvh_scene, scatter_scene
        return deep_bvh_scene(1_000_000) if scene.endswith("deep_bvh_1M") else scatter_scene(1_000_000)
    return rpt.World.from_path(rpt.fixture(scene + ".glb"))


def load_traffic(hip, workload):
    """profiles/traffic_<workload>.json, but only when it was measured on exactly the kernel sources of the LOADED library
    (tools/source_fingerprint.py; PMC counters cannot be collected from inside this process) -> (json or None, note)"""
    tpath = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_fingerprint import fingerprint
        with open(tpath) as f:
            tj = json.load(f)
        fp = hip.build_fingerprint()             # of the LOADED library, not of whatever the source tree holds now
        if fp != fingerprint():
            print(f"bench: librpt_hip.so was built from other sources ({fp}) than the tree holds ({fingerprint()}): run make", file=sys.stderr)
        if tj.get("workload") == workload and tj.get("source_fingerprint") == fp:
            return tj, f"profiles/traffic_{workload}.json@{fp}"
        return None, f"profiles/traffic_{workload}.json is from other kernel sources ({tj.get('source_fingerprint')} != {fp}): not reported"
    except Exception as e:                       # noqa: BLE001
        return None, f"profiles/traffic_{workload}.json unreadable: {e}"


# kernels whose time the library's per-stage HIP events attribute to one stage (rpt_stats.kernel_ms): the any-hit walk and the
# dense pass that adds the unoccluded NEE terms are timed together as "shadow"
STAGE_KERNELS = {"shadow": ("shadow", "shadow_resolve")}


def stage_roofline(hip, workload, s0, s1, steps, elapsed_s, cus, clock_mhz, pipeline_bytes, share=1.0):
    """(roofline of the dominant stage, whole-batch traffic / VALU figures) from the HIP events recorded in this run and the kept
    PMC passes of the same kernel sources.  `share`: the slots one launch of this run covers over the slots a launch of the kept PMC passes
    covered (whole-image launches of 32-sample batches on one GPU): the part of the image this rank renders x spp_per_step / 32 — every
    per-launch counter figure is scaled by it (and says so when it is not 1)."""
    kms = {k: s1["kernel_ms"][k] - s0["kernel_ms"][k] for k in s1["kernel_ms"]}
    klaunch = {k: s1["kernel_launches"][k] - s0["kernel_launches"][k] for k in s1["kernel_launches"]}
    dominant = max(kms, key=lambda k: kms[k])
    if kms[dominant] <= 0:
        return None, None
    avg_ms = kms[dominant] / max(klaunch[dominant], 1)
    unit_key, bytes_per_unit = STAGE_MODEL[dominant]
    units = (s1[unit_key] - s0[unit_key]) / max(klaunch[dominant], 1)
    achieved = bytes_per_unit * units / (avg_ms * 1e-3) / 1e9
    tj, traffic_source = load_traffic(hip, workload)
    stages = tj.get("stages", {}) if tj else {}
    # the stage's figures = the sum over every kernel its avg_launch_ms covers
    covered = [k for k in STAGE_KERNELS.get(dominant, (dominant,)) if k in stages]
    traffic = valu = ta = None
    if dominant in covered:
        traffic = int(sum(stages[k]["hbm_bytes_per_launch"] for k in covered) * share)
        if all("valu" in stages[k] for k in covered):
            insts = sum(stages[k]["valu"]["wave_instructions_per_launch"] for k in covered)
            valu = {"wave_instructions_per_launch": int(insts * share),
                    "lane_utilisation": round(sum(stages[k]["valu"]["lane_utilisation"] * stages[k]["valu"]["wave_instructions_per_launch"]
                                                  for k in covered) / max(insts, 1), 4)}
            for key in ("class_mix", "issue_cycles_per_wave_instruction"):
                if key in stages[dominant]["valu"]:
                    valu[key] = stages[dominant]["valu"][key]
        ta = stages[dominant].get("ta")
    algorithmic = bytes_per_unit * units
    roofline = {"bound": "hbm", "kernel": "k_" + dominant, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "traffic_over_algorithmic": round(traffic / algorithmic, 4) if traffic and algorithmic > 0 else None,
                "traffic_kernels": ["k_" + k for k in covered] or None, "traffic_source": traffic_source,
                "avg_launch_ms": round(avg_ms, 5), "launches": int(klaunch[dominant]),
                "units_per_launch": round(units, 1), "units_are": unit_key, "algorithmic_bytes_per_unit": bytes_per_unit,
                "stage_ms": {k: round(v, 3) for k, v in kms.items()},
                "stage_launches": {k: int(v) for k, v in klaunch.items()}}
    if share != 1.0 and traffic is not None:
        roofline["traffic_scaled_by"] = round(share, 6)      # whole-image 32-sample PMC launches -> the slots of this rank's launches
    if ta:
        # the streamed global-memory walks are bound by the CU's texture-address unit, not by HBM: its busy share from the kept TA pass
        roofline["ta"] = ta
    # `bound` is the contract's label (achieved algorithmic bytes / s against the HBM peak); what the kept counters say limits the stage is named beside it,
    # so that `frac` is not read as headroom that does not exist
    if valu or ta:
        busy = (ta or {}).get("busy_frac")
        roofline["limiter_measured"] = ("texture-address unit (global-memory walk: TA busy %.0f %%)" % (100 * busy) if busy and busy > 0.5 else
                                        "VALU issue at %.0f %% live lanes (see valu / pipeline_roofline.valu.issue_frac), not HBM" % (100 * valu["lane_utilisation"]) if valu else None)
    simds = cus * 4
    if valu:
        # not an HBM kernel: the VALU issue figures of the kept SQ pass (same fingerprint rule as `traffic`), and from them
        # and THIS run's launch time the SIMD time per wave-instruction
        n_inst = max(valu["wave_instructions_per_launch"], 1)
        roofline["valu"] = dict(valu, simd_ns_per_wave_instruction=round(avg_ms * 1e6 * simds / n_inst, 4),
                                simd_cycles_per_wave_instruction=round(avg_ms * 1e3 * clock_mhz * simds / n_inst, 3),
                                simds=simds, clock_mhz=clock_mhz,
                                note="the traversal mix issues in ~2.7 SIMD cycles per wave64 instruction (DESIGN.md 4)")
    # whole batch: measured HBM bytes of ALL kernels (FETCH x 2 + WRITE per launch x this run's launches) against SURVEY.md 8d's
    # algorithmic bytes, and the share of the SIMD issue cycles of the timed region the VALU instructions of all kernels need
    whole = None
    if tj:
        launches_of = dict(klaunch)
        launches_of["shadow_resolve"] = klaunch.get("shadow", 0)
        tot_bytes, tot_insts, tot_issue, covered_all, missing = 0.0, 0.0, 0.0, [], []
        for st, rec in stages.items():
            n = launches_of.get(st, 0)
            if n <= 0:
                continue
            tot_bytes += rec["hbm_bytes_per_launch"] * n * share
            if "valu" in rec:
                ni = rec["valu"]["wave_instructions_per_launch"] * n * share
                tot_insts += ni
                # issue cycles per wave-instruction of THIS kernel's measured class mix (tools/valu_from_pmc.py) where the kept
                # SQ passes carry one, the fastest class's figure otherwise
                tot_issue += ni * rec["valu"].get("issue_cycles_per_wave_instruction", VALU_ISSUE_CYCLES_FLOOR)
            covered_all.append(st)
        for st, n in klaunch.items():
            if n > 0 and st not in stages:
                missing.append(st)
        avail = simds * clock_mhz * 1e6 * elapsed_s
        mixed = any("issue_cycles_per_wave_instruction" in rec.get("valu", {}) for rec in stages.values())
        whole = {"traffic": int(tot_bytes / max(steps, 1)), "traffic_over_algorithmic": round(tot_bytes / max(pipeline_bytes, 1), 4),
                 "traffic_is": "HBM bytes per batch, all kernels: (2 x FETCH_SIZE + WRITE_SIZE) per launch of the kept PMC passes x the launches of this run"
                               + ("" if share == 1.0 else f" x the slots of this rank's launches over a whole-image 32-sample launch ({share:.4f})"),
                 "stages_covered": covered_all, "stages_without_counters": missing,
                 "valu": {"wave_instructions_per_batch": int(tot_insts / max(steps, 1)),
                          "simd_cycles_per_wave_instruction": round(avail / tot_insts, 3) if tot_insts else None,
                          "issue_frac": round(tot_issue / avail, 4) if tot_insts else None,
                          "issue_frac_is": ("sum over kernels of VALU wave-instructions x the issue cycles of that kernel's measured instruction-class mix "
                                            "(SQ_INSTS_VALU_* passes; fma-class 2.1, other classes 3.45 SIMD cycles, a third of 'other' hides: profiles/r04_valu_pipes.txt)"
                                            if mixed else
                                            f"all kernels' VALU wave-instructions x {VALU_ISSUE_CYCLES_FLOOR} cycles (the fastest class, v_fma_f32; "
                                            "compares / min / max / integer take ~4)") + " / (SIMDs x clock x timed seconds)"}}
    return roofline, whole


def parity_windows(world, cfg, seeds, image, image_spp, want_spp, W, H, budget_s=40.0):
    """windows of an image the GPU has just rendered vs the CPU oracle at the same sample count (same scene buffers, config, seeds):
    accumulators must be equal BIT FOR BIT (the sum order over all samples is part of the result, kernels/src/lib.rs:225-226)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_ffi import Oracle
    orc = Oracle("rpt_math")
    osc = orc.scene(world)
    ww, wh = min(48, W), min(40, H)
    # centre, across a 64 x 64 tile seam (x = 64 k: two ranks' tiles when N > 1), bottom-right corner
    seam_x, seam_y = 64 * max(1, W // 192), 64 * max(1, H // 320)         # a tile corner away from the centre (1024^2: 320, 192)
    clampx, clampy = (lambda v: max(0, min(W - ww, v))), (lambda v: max(0, min(H - wh, v)))
    rects = [(clampx((W - ww) // 2), clampy((H - wh) // 2)), (clampx(seam_x - ww // 2), clampy(seam_y - wh // 2)), (W - ww, H - wh)]
    t_par = time.perf_counter()
    n_win, bitwise, num, den, worst = 0, True, 0.0, 0.0, 0
    for (x0, y0) in rects:
        if n_win >= 2 and time.perf_counter() - t_par > budget_s:
            break
        ref, _, _ = orc.trace_cpu(cfg, osc, seeds, want_spp, rect=(x0, y0, x0 + ww, y0 + wh), threads=usable_cores())
        a = image[y0:y0 + wh, x0:x0 + ww]
        b = ref[y0:y0 + wh, x0:x0 + ww]
        same = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
        bitwise = bitwise and same
        worst = max(worst, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
        num += float(((a[..., :3].astype(np.float64) - b[..., :3]) ** 2).sum())
        den += float((b[..., :3].astype(np.float64) ** 2).sum())
        n_win += 1
    parity = {"windows": n_win, "window_px": [ww, wh], "spp": int(want_spp), "image_spp": int(image_spp), "bitwise": bitwise,
              "rel_l2": (num / den) ** 0.5 if den > 0 else 0.0, "differing_words": worst,
              "against": "oracle/rpt_oracle.cpp trace_cpu(rect) at the same spp, config, seeds", "seconds": round(time.perf_counter() - t_par, 2)}
    if image_spp != want_spp:
        parity["bitwise"] = False
        parity["error"] = f"image carries {image_spp} spp, expected {want_spp}"
    return parity


def parity_ok(parity):
    return parity is None or (parity["bitwise"] and parity["rel_l2"] <= 1e-4)


def workload_label(scene, W, H, steps, spp_per_step, total_spp, cfg):
    kind, _, name = scene.rpartition(":")
    what = {"": name + ".glb", "synthetic-textures": name + ".glb + synthetic textures", "procedural": "procedural:" + name}[kind]
    return (f"{what} {W}x{H}, {steps}x{spp_per_step} spp (config total {total_spp}), "
            f"nee={cfg.nee}, bounces {cfg.min_bounces}/{cfg.max_bounces}")


def measure_single_gpu_workload(rpt, hip, name, steps, warmup, spp_per_step, device_index, cus, clock_mhz, parity=True, measure_startup=False):
    """One more BASELINE workload on ONE GPU, on a fresh context, after (never inside) the headline's timed loop: the same step
    (rpt_render_async of one batch), the same bracketing, the same roofline / parity objects as the headline line."""
    import numpy as np
    import torch
    scene, W, H, total_spp, over = WORKLOADS[name]
    world = build_world(rpt, scene)
    cfg = rpt.default_config(W, H, **over)
    seeds = rpt.blue_noise_seeds(W, H)
    startup = None
    r = hip.Renderer(device_index)
    try:
        if measure_startup:
            # "Startup time (GPU)" of the reference's own bench (benches/benchmark.rs:11-13: trace_gpu(scene, 0 samples) = load the scene, build the BVH
            # and the light table, create the buffers) for this scene, scene-preparation steps on the device: BVH (rpt_bvh_build_gpu) + light table
            # (rpt_light_table_build_gpu) from the raw triangles, then rpt_upload_scene (upload-time derivations + the shadow-order probe), set_config,
            # reset; the first batch (which also allocates the path state) is reported beside it, with the steady batch.  File parsing has no counterpart here:
            # the stand-in is generated.
            torch.cuda.synchronize()
            v = np.ascontiguousarray(world.per_vertex["vertex"], np.float32).reshape(-1, 4)
            tiny_v = np.ascontiguousarray(np.random.default_rng(0).random((48, 4)), np.float32)     # (code objects loaded, as any second scene of a session finds them)
            tiny_t = np.zeros(16, world.indices.dtype)
            tiny_t["v0"], tiny_t["v1"], tiny_t["v2"] = np.arange(0, 48, 3), np.arange(1, 48, 3), np.arange(2, 48, 3)
            hip.bvh_build_gpu(tiny_v, tiny_t)
            hip.light_table_build_gpu(tiny_v, tiny_t, world.materials)
            t_s = time.perf_counter()
            nodes, tris, bvh_dev_ms = hip.bvh_build_gpu(v, world.indices)
            t_b = time.perf_counter()
            table, n_em, lt_ms = hip.light_table_build_gpu(v, tris, world.materials)
            t_l = time.perf_counter()
            built = type(world).__new__(type(world))
            built.__dict__.update(world.__dict__)
            built.nodes, built.indices, built.light_pick = nodes, tris, table
            r.upload_scene(built)
            t_u = time.perf_counter()
            r.set_config(cfg)
            r.reset(seeds)
            t_c = time.perf_counter()
            r.render(spp_per_step)
            t_f = time.perf_counter()
            startup = {"as": "benches/benchmark.rs:11-13 'Startup time (GPU)' = trace_gpu(scene, 0 samples), on the stand-in (no file to parse)",
                       "triangles": int(len(world.indices)), "nodes": int(len(nodes)), "light_table_entries": int(len(table)),
                       "bvh_build_gpu_ms": round((t_b - t_s) * 1e3, 2), "bvh_build_device_ms": round(bvh_dev_ms, 2),
                       "light_table_gpu_ms": round((t_l - t_b) * 1e3, 2), "light_table_breakdown_ms": {k: round(x, 2) for k, x in lt_ms.items()},
                       "upload_scene_ms": round((t_u - t_l) * 1e3, 2), "upload_scene_shadow_order_probe_ms": round(r.shadow_order()["probe_ms"], 2),
                       "set_config_reset_ms": round((t_c - t_u) * 1e3, 2),
                       "startup_ms": round((t_c - t_s) * 1e3, 2), "first_batch_ms": round((t_f - t_c) * 1e3, 2),
                       "checked_by": "tests/test_gpu_bvh_build.py, tests/test_gpu_light_table.py: both builds equal the sequential builders bit for bit"}
        # (the measured loop and its parity check run on the scene as the host built it: the triangles above were already in BVH order,
        # so that build is another — equally valid — tree)
        r.upload_scene(world)
        r.set_config(cfg)
        r.reset(seeds)
        for _ in range(2):                                           # set-up: touch every page of the path state, clocks up
            r.render(spp_per_step)
        r.reset(seeds)
        for _ in range(warmup):
            r.render_async(spp_per_step)
        r.wait()
        torch.cuda.synchronize()
        s0 = r.stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render_async(spp_per_step)
        r.wait()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        s1 = r.stats()
        # shadow rays WALKED on the device (what Mrays/s counts); rpt_stats.shadow_rays keeps the reference's count — one per NEE evaluation —
        # of which shadow_rays_elided were not walked: their term is zero whatever the walk finds (k_shade.h)
        n_ext, n_shadow = s1["extension_rays"] - s0["extension_rays"], s1["shadow_rays_traced"] - s0["shadow_rays_traced"]
        n_elided = s1["shadow_rays_elided"] - s0["shadow_rays_elided"]
        n_samples, n_sky = s1["samples"] - s0["samples"], s1["sky_evals"] - s0["sky_evals"]
        n_mis = (n_shadow + n_elided) if cfg.nee == 1 else 0                    # the MIS carry is written per NEE evaluation
        pipeline_bytes = algorithmic_bytes(n_ext, n_shadow, n_mis, n_samples)
        roofline, whole = stage_roofline(hip, name, s0, s1, steps, elapsed, cus, clock_mhz, pipeline_bytes)
        pgbs = pipeline_bytes / elapsed / 1e9
        order, last_order = shadow_order_label(r), last_bounce_label(r)
        par = None
        if parity:
            image, image_spp = r.read_accum()
            par = parity_windows(world, cfg, seeds, image, image_spp, spp_per_step * (warmup + steps), W, H, budget_s=15.0)
    finally:
        r.close()
    pipeline = {"bound": "hbm", "achieved": round(pgbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pgbs / HBM_PEAK_GBS, 6)}
    if whole:
        pipeline.update({k: whole[k] for k in ("traffic", "traffic_over_algorithmic", "stages_without_counters")})
        pipeline["valu"] = whole["valu"]
    # (the headline carries the explanatory strings once; the line stays well under the driver's 24 000-character tail)
    if roofline:
        roofline.get("valu", {}).pop("note", None)
    if par:
        par.pop("against", None)
    if "valu" in pipeline:
        pipeline["valu"].pop("issue_frac_is", None)
    if startup:
        # (the first batch also allocates the path state; beside it the batch of the timed loop, so that nobody reads the batch itself as a start-up cost)
        startup["steady_batch_ms"] = round(elapsed / steps * 1e3, 2)
    return {"value": round((n_ext + n_shadow) / elapsed / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(elapsed / steps * 1e3, 4),
            "value_as_the_reference_counts": round((n_ext + n_shadow + n_elided) / elapsed / 1e6, 3),   # + the shadow rays the reference traces and this build proves irrelevant
            "steps": steps, "warmup": warmup, "samples_per_s": round(n_samples / elapsed, 1),
            "data": data_label(scene),
            "config": {"workload": workload_label(scene, W, H, steps, spp_per_step, total_spp, cfg), "spp_per_step": spp_per_step,
                       "shadow_order": order, "last_bounce_order": last_order},
            "rays": {"extension": int(n_ext), "shadow": int(n_shadow), "shadow_elided": int(n_elided), "sky_evals": int(n_sky),
                     "per_sample": round((n_ext + n_shadow) / max(n_samples, 1), 4),
                     "per_sample_as_the_reference_counts": round((n_ext + n_shadow + n_elided) / max(n_samples, 1), 4)},
            "roofline": roofline, "pipeline_roofline": pipeline, "parity_check": par, **({"startup": startup} if startup else {})}


def shadow_order_label(r):
    """Which of the two bit-exact visiting orders the scene's any-hit walks use (csrc/shadow_order.h decides at upload), and the probe's numbers."""
    so = r.shadow_order()
    return {"order": "fixed, more opaque child first" if so["fixed"] else "near child first", "probe_rays": so["probe_rays"],
            "probe_node_visits_near_first": round(so["visits_near"], 2), "probe_node_visits_fixed": round(so["visits_fixed"], 2)}


def last_bounce_label(r):
    """How the last extension rays of a batch without NEE are walked on this scene (rpt_last_bounce_order; every mode gives the same image)."""
    lo = r.last_bounce_order()
    return {"mode": lo["mode_is"], "emissive_triangles": lo["emissive_triangles"], "probe_rays": lo["probe_rays"],
            "probe_node_visits": {k: round(v, 2) for k, v in lo["probe_node_visits"].items()}}


HANG_HINTS = ("hints: a communicator bring-up that never returns is usually fabric / IPC configuration — run once with NCCL_DEBUG=INFO (NCCL_DEBUG=WARN is "
              "set by this launcher), check HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC is the only mode the host driver supports), `rocm-smi --showtopo` for the xGMI "
              "links, and that every rank sees its own device (HIP_VISIBLE_DEVICES); --launch-timeout raises the watchdog, --driver multi starts the "
              "one-process driver (rpt_multi_*: ncclCommInitAll) instead of one process per GPU")


def run_child(cmd, env, timeout_s, label):
    """Run one child process under a watchdog: its own session (= its own process group), stdout collected, stderr passed through and its
    last 50 lines kept.  On expiry the child's WHOLE process group — the group this function created, nothing found by name — is killed.
    -> (return code or None when killed, stdout lines, last stderr lines, timed out)"""
    import collections
    import signal
    import subprocess
    import threading
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, text=True, errors="replace")
    tail, out_lines = collections.deque(maxlen=50), []

    def pump_err():
        for line in p.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            tail.append(line)

    def pump_out():
        for line in p.stdout:
            out_lines.append(line)
    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in threads:
        t.start()
    timed_out = False
    try:
        rc = p.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        timed_out, rc = True, None
        try:
            os.killpg(p.pid, signal.SIGKILL)              # the session leader's pid is the group id: exactly the processes this call started
        except ProcessLookupError:
            pass
        p.wait()
    for t in threads:
        t.join(timeout=10)
    if timed_out:
        print(f"bench: {label} did not finish within {timeout_s:.0f} s: its process group was killed.  Last lines of its stderr:", file=sys.stderr)
        for line in tail:
            sys.stderr.write("    | " + line)
        print("bench: " + HANG_HINTS, file=sys.stderr)
        sys.stderr.flush()
    return rc, out_lines, list(tail), timed_out


def launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment (the driver's plain command; a caller of src/trace.rs:136-224
    does not bring one either): this process — which has not imported torch nor touched HIP — becomes the launcher.  Every attempt is a CHILD
    process under a watchdog (--launch-timeout), never an exec:
      1. one process per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <argv>` on a free port of 127.0.0.1
         (rpt_comm_init = ncclCommInitRank per rank);
      2. only if that failed or hung (or with --driver multi): ONE process driving all N GPUs through rpt_multi_* (ncclCommInitAll — a different
         RCCL bring-up path); its line says so in config.driver / config.fallback_from.
    The child's one JSON line and return code are relayed."""
    import socket
    n = args.gpus
    base_env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base_env.setdefault("OMP_NUM_THREADS", "1")
    base_env.setdefault("NCCL_DEBUG", "WARN")
    me = os.path.abspath(__file__)
    why = None
    if args.driver in ("auto", "ranks"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), me] + list(argv)
        print("bench: no launcher in the environment: starting " + " ".join(cmd[1:8]) + f" as a child process (watchdog {args.launch_timeout:.0f} s)", file=sys.stderr)
        sys.stderr.flush()
        rc, out, tail, timed_out = run_child(cmd, base_env, args.launch_timeout, f"the {n}-rank run (one process per GPU)")
        lines = [ln for ln in out if ln.startswith("{")]
        if rc == 0 and lines:
            sys.stdout.write(lines[-1])
            sys.stdout.flush()
            return 0
        why = f"the per-process driver {'hung (killed after %.0f s)' % args.launch_timeout if timed_out else 'exited with code %s' % rc}"
        if args.driver == "ranks":
            print(f"bench: {why}; --driver ranks: no second attempt", file=sys.stderr)
            return rc if rc else 1
        print(f"bench: {why}: second attempt with ONE process driving all {n} GPUs (rpt_multi_*, ncclCommInitAll)", file=sys.stderr)
    env = dict(base_env, RPT_BENCH_CHILD="multi")
    if why:
        env["RPT_BENCH_FALLBACK_FROM"] = why
    rc, out, tail, timed_out = run_child([sys.executable, me] + list(argv), env, args.launch_timeout, f"the one-process {n}-GPU run (rpt_multi_*)")
    lines = [ln for ln in out if ln.startswith("{")]
    if rc == 0 and lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
        return 0
    print(f"bench: the one-process driver {'hung' if timed_out else 'exited with code %s' % rc}: no scaling line", file=sys.stderr)
    return rc if rc else 1


def run_multi_driver(args):
    """--driver multi: ONE process drives all N GPUs through rpt_multi_* (ncclCommInitAll inside rpt_multi_create) — the shape of the reference's single
    render thread (src/trace.rs:136-224).  Same step (one batch on every GPU + the batch's one gather), same bracketing, same JSON line; no
    torch.distributed (there is one process).  config.driver says which driver produced the line."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)                                           # RCCL's banner goes to C stdout: keep the JSON line's descriptor aside
    os.environ.setdefault("RPT_STAGE_TIMING", "2")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np                                      # noqa: F401
    import torch
    rpt = importlib.import_module("rust-path-tracer_amd")
    hip = importlib.import_module("rust-path-tracer_amd.hip")
    n = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    shared = bool(args.rehearsal)
    if not shared and os.environ.get("RPT_RCCL_LIBRARY"):
        raise SystemExit("bench.py: RPT_RCCL_LIBRARY is set — that override loads a stand-in for RCCL (tests/fake_rccl); nothing measured with it is a measurement")
    if not shared and torch.cuda.device_count() < n:
        raise SystemExit(f"--driver multi --gpus {n}: only {torch.cuda.device_count()} HIP devices visible")
    scene, W, H, total_spp, over = WORKLOADS[args.workload]
    world = build_world(rpt, scene)
    cfg = rpt.default_config(W, H, **over)
    seeds = rpt.blue_noise_seeds(W, H)
    m = hip.MultiRenderer([0] * n if shared else list(range(n)), allow_shared_device=shared)
    m.upload_scene(world)
    m.set_config(cfg)
    m.reset(seeds)
    for _ in range(2):                                      # set-up: pages touched, clocks up
        m.render(args.spp_per_step)
    m.wait()
    m.reset(seeds)

    def sync_all():
        m.wait()
        for d in ([0] if shared else range(n)):
            torch.cuda.synchronize(d)
    for _ in range(args.warmup):
        m.render(args.spp_per_step)
    sync_all()
    r0 = m.rank_view(0)
    s0, k0 = m.stats(), r0.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.render(args.spp_per_step)                          # one batch on every GPU + the batch's single gather, enqueued
    sync_all()                                               # the last batch's image is complete on rank 0 inside the timed region
    elapsed = time.perf_counter() - t0
    s1, k1 = m.stats(), r0.stats()
    n_ext, n_samples = s1["extension_rays"] - s0["extension_rays"], s1["samples"] - s0["samples"]
    n_elided = s1["shadow_rays_elided"] - s0["shadow_rays_elided"]
    n_shadow = (s1["shadow_rays"] - s0["shadow_rays"]) - n_elided
    n_sky = s1["sky_evals"] - s0["sky_evals"]
    n_mis = (n_shadow + n_elided) if cfg.nee == 1 else 0
    cus, clock_mhz = hip.device_info(0)
    pipeline_bytes = algorithmic_bytes(n_ext, n_shadow, n_mis, n_samples)
    share = r0.local_pixels() / float(W * H)
    roofline, whole = stage_roofline(hip, args.workload, k0, k1, args.steps, elapsed, cus, clock_mhz, pipeline_bytes * share, share * args.spp_per_step / 32.0)
    parity = None
    if not args.no_parity_check:
        image, image_spp = m.read_accum()
        parity = parity_windows(world, cfg, seeds, image, image_spp, args.spp_per_step * (args.warmup + args.steps), W, H)
    order, last_order = shadow_order_label(r0), last_bounce_label(r0)
    out = {}
    if args.rehearsal:
        out["rehearsal"] = "NOT A MEASUREMENT: all ranks share one GPU (RPT_MULTI_ALLOW_SHARED_DEVICE: device-to-device copies stand in for RCCL)"
    pgbs = pipeline_bytes / elapsed / 1e9
    out.update({
        "metric": "Mrays/s", "value": round((n_ext + n_shadow) / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": data_label(scene) + ("" if ":" in scene else " + blue-noise seeds; no synthetic geometry"),
        "config": {"workload": workload_label(scene, W, H, args.steps, args.spp_per_step, total_spp, cfg), "spp_per_step": args.spp_per_step,
                   "tiles": "64x64 round-robin", "kernel_sources": hip.build_fingerprint(),
                   "driver": "multi: ONE process drives all GPUs through rpt_multi_* (ncclCommInitAll)",
                   "fallback_from": os.environ.get("RPT_BENCH_FALLBACK_FROM"),
                   "gather": "rccl-c-abi" if not shared else "device copies (shared-device test aid)", "collective_library": hip.comm_library() or None,
                   "rpt_comm_world": list(r0.comm_world()), "shadow_order": order, "last_bounce_order": last_order},
        "samples_per_s": round(n_samples / elapsed, 1),
        "value_as_the_reference_counts": round((n_ext + n_shadow + n_elided) / elapsed / 1e6, 3),
        "rays": {"extension": int(n_ext), "shadow": int(n_shadow), "shadow_elided": int(n_elided), "sky_evals": int(n_sky),
                 "per_sample": round((n_ext + n_shadow) / max(n_samples, 1), 4)},
        "roofline": roofline,
        "pipeline_roofline": {"bound": "hbm", "achieved": round(pgbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pgbs / HBM_PEAK_GBS, 6),
                              "formula": "128*N_ext + 96*N_shadow + 128*N_mis + 40*samples (SURVEY.md 8d)"},
        "cpu_baseline": None, "parity_check": parity, "readback": None,
    })
    if whole:
        out["pipeline_roofline"].update({k: whole[k] for k in ("traffic", "traffic_over_algorithmic", "traffic_is", "stages_without_counters")})
    m.close()
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if not parity_ok(parity):
        print("bench: PARITY CHECK FAILED: " + json.dumps(parity), file=sys.stderr)
        raise SystemExit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3, help="untimed batches first (clocks ramp up over the first ~50 ms)")
    ap.add_argument("--workload", default="darkcornell", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=None,
                    help="samples per batch; default 32 x N (at most 256): the reference's sync_rate of 32 (src/trace.rs:75) at N = 1, and at N GPUs "
                         "as many samples of a pixel in flight as make a rank's launches cover the slots the whole image covers at 32")
    ap.add_argument("--driver", default="auto", choices=["auto", "ranks", "multi"],
                    help="N > 1 without a launcher: ranks = one process per GPU under torch.distributed.run; multi = ONE process through rpt_multi_* "
                         "(ncclCommInitAll); auto = ranks, and multi only if that failed or hung")
    ap.add_argument("--launch-timeout", type=float, default=600.0, help="watchdog (seconds) around each self-launched child")
    ap.add_argument("--rehearsal-fail-ranks", action="store_true",
                    help="test aid, with --rehearsal only: every rank of the one-process-per-GPU driver exits at once (code 5), so that the launcher's second attempt — "
                         "the one-process driver — is exercised end to end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo (host-staged gather) only exists to exercise the N>1 logic on a box with one GPU")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="testing aid for --dist-backend gloo")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "torch"],
                    help="rccl: the per-batch gather runs inside librpt_hip.so (C ABI, RCCL); torch: tiles.Gatherer over torch.distributed")
    ap.add_argument("--with-gather", action="store_true", help="N = 1 only: still run the per-batch gather (a 1-rank communicator), to exercise that path")
    ap.add_argument("--rehearsal", action="store_true",
                    help="dress rehearsal of the N > 1 path on a box with ONE GPU: N processes on device 0, torch.distributed over gloo, the "
                         "library's gather over the test stand-in for RCCL (RPT_RCCL_LIBRARY).  Exercises every line the scaling run executes; "
                         "the JSON line is marked and its value is not a measurement")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the untimed comparison of windows of the rendered image with the CPU oracle")
    ap.add_argument("--extra-workloads", default="darkcornell_mis,veachmis,pbrtest,pbrtest_textured,deepbvh",
                    help="N = 1, headline workload only: the other single-GPU BASELINE workloads, measured AFTER the headline's timed loop on fresh "
                         "contexts and reported under \"workloads\" (value, roofline, pipeline_roofline, parity_check each); a failed parity check fails the bench")
    ap.add_argument("--no-extra-workloads", action="store_true")
    ap.add_argument("--extra-steps", type=int, default=4)
    ap.add_argument("--extra-warmup", type=int, default=1)
    ap.add_argument("--no-readback", action="store_true", help="skip the extra render -> read_accum loop (reference loop shape, src/trace.rs:182-204)")
    args = ap.parse_args()

    if args.spp_per_step is None:
        args.spp_per_step = min(256, 32 * max(1, args.gpus))
    # `python bench.py --gpus N` with N > 1 and no launcher: this process becomes the launcher of watchdog-guarded CHILD processes (launch()).
    in_launcher = int(os.environ.get("WORLD_SIZE", "1") or "1") > 1      # (an inherited WORLD_SIZE=1 is no launcher either)
    if os.environ.get("RPT_BENCH_CHILD") == "multi" and args.gpus > 1 and not in_launcher:
        return run_multi_driver(args)
    if args.rehearsal_fail_ranks and args.rehearsal and in_launcher:
        print(f"bench: rank {os.environ.get('RANK')}: --rehearsal-fail-ranks: failing on purpose", file=sys.stderr)
        raise SystemExit(5)
    if args.gpus > 1 and not in_launcher:
        raise SystemExit(launch(args, sys.argv[1:]))

    # Rank 0 prints ONE JSON line on stdout.  Libraries print there too (RCCL writes a version banner through C stdio when
    # a communicator is created, and it is flushed at exit — after the JSON line), so for the whole run file descriptor 1
    # points at stderr and the JSON line goes to the real stdout, kept aside here.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # HIP events on the render stream: after every stage kernel at N = 1 (per-stage breakdown), only around the
    # traversal kernel (the dominant one, which the roofline needs) at N > 1, where a 16-kernel batch lasts ~1.3 ms
    # and 20 event records per batch cost 7 % of it
    os.environ.setdefault("RPT_STAGE_TIMING", "1" if int(os.environ.get("WORLD_SIZE", "1")) <= 1 else "2")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    rpt = importlib.import_module("rust-path-tracer_amd")
    hip = importlib.import_module("rust-path-tracer_amd.hip")
    tiles = importlib.import_module("rust-path-tracer_amd.tiles")

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.rehearsal:
        args.dist_backend, args.all_ranks_on_device0 = "gloo", True
        if not os.environ.get("RPT_RCCL_LIBRARY"):
            raise SystemExit("--rehearsal needs RPT_RCCL_LIBRARY=tests/fake_rccl/librccl_fake.so (real RCCL refuses two ranks on one device)")
    elif os.environ.get("RPT_RCCL_LIBRARY"):
        raise SystemExit("bench.py: RPT_RCCL_LIBRARY is set — that override loads a stand-in for RCCL (tests/fake_rccl); nothing measured with it is a measurement")
    if args.all_ranks_on_device0:
        local_rank = 0
    elif world_size > 1 and torch.cuda.device_count() == 1:
        local_rank = 0            # a launcher that exposes ONE GPU per process (HIP_VISIBLE_DEVICES per rank): this rank's GPU is its device 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device(device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)

    scene, W, H, total_spp, over = WORKLOADS[args.workload]
    world = build_world(rpt, scene)
    cfg = rpt.default_config(W, H, **over)
    seeds = rpt.blue_noise_seeds(W, H)

    r = hip.Renderer(local_rank, rank=rank, world_size=world_size)
    r.upload_scene(world)
    r.set_config(cfg)
    r.reset(seeds)
    # set-up, not measurement: two throw-away batches touch every page of the path state (6.7 GB at 1024^2) and bring
    # the clocks up, then the accumulators start over from zero samples
    for _ in range(2):
        r.render(args.spp_per_step)
    r.reset(seeds)
    image = None
    comm_device = device if args.dist_backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The single collective per sample batch: inside the library, behind the C ABI (rpt_comm_init = RCCL ncclCommInitRank,
    # rpt_gather_async = grouped ncclSend/ncclRecv to rank 0 on the library's second HIP stream, root un-tile with a map
    # built once): stream-ordered after the batch, nothing on the host per step, batch k+1 renders while the blocks of
    # batch k travel.  torch.distributed only carries the 128-byte unique id, the barrier and the final statistics.
    # If RCCL cannot be initialised from the library the bench FAILS (non-zero exit): a number from another gather would not
    # be the product's.  --gather torch (tiles.Gatherer over torch.distributed, the round-1 path) exists only as an explicit
    # choice and says so in `config.gather`.
    gather_impl = None
    if world_size > 1 or args.with_gather:
        want = args.gather if (args.dist_backend == "nccl" or args.rehearsal) else "torch"
        if want == "rccl":
            gather_note = None
            uid = None
            if rank == 0:
                try:
                    uid = hip.comm_unique_id()
                except Exception as e:                               # noqa: BLE001 — reported below, on every rank
                    gather_note = f"{type(e).__name__}: {e}"
            ids = [uid]
            if world_size > 1:                                       # every rank takes part, whatever rank 0 got
                dist.broadcast_object_list(ids, src=0, device=torch.device(comm_device))
            ok = torch.tensor([0.0], device=comm_device)
            if ids[0] is not None:
                try:
                    r.comm_init(ids[0], rank, world_size)
                    if r.comm_world() != (rank, world_size):
                        raise RuntimeError(f"communicator reports {r.comm_world()}, expected {(rank, world_size)}")
                    ok = torch.tensor([1.0], device=comm_device)
                except Exception as e:                               # noqa: BLE001
                    gather_note = f"{type(e).__name__}: {e}"
            if world_size > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok[0]) < 1.0:
                print(f"bench: rank {rank}: the library's RCCL gather could not be set up ({gather_note or 'another rank failed'}); "
                      "no fallback — `--gather torch` is a different, explicitly chosen code path", file=sys.stderr)
                if world_size > 1:
                    dist.destroy_process_group()
                raise SystemExit(3)
            gather_impl = "rccl-c-abi"
        else:
            gather_impl = "torch.distributed (explicit --gather torch / --dist-backend gloo: NOT the product's gather)"
    use_lib_gather = gather_impl == "rccl-c-abi"
    comm_world_seen = list(r.comm_world()) if use_lib_gather else None
    gatherer = lib_stream = local_block = staged = None
    if gather_impl and not use_lib_gather:
        local_block = tiles.device_block_as_tensor(r, device)
        image = torch.zeros((H, W, 4), dtype=torch.float32, device=device) if rank == 0 else None
        gatherer = tiles.Gatherer(W, H, comm_device)
        staged = torch.zeros((world_size, gatherer.stride, 4), dtype=torch.float32, device=device) \
            if (rank == 0 and comm_device == "cpu") else None
        lib_stream = torch.cuda.ExternalStream(r.stream_ptr(), device=device)

    def finish_gather():
        """(torch path) complete the gather of the previous batch; rank 0 un-tiles it into the full image."""
        recv = gatherer.end()                                        # (stream-level wait under lib_stream)
        if rank == 0 and recv is not None:
            if staged is not None:
                staged.copy_(recv)
                recv = staged
            r.untile(recv.data_ptr(), image.data_ptr(), gatherer.stride)     # launch only, on the same stream

    def step():
        r.render_async(args.spp_per_step)
        if use_lib_gather:
            r.gather_async()                                         # stream-ordered; returns at once
        elif gatherer is not None:
            with torch.cuda.stream(lib_stream):
                finish_gather()                                      # batch k-1 travelled while batch k rendered
                gatherer.begin(local_block)

    def drain():
        if use_lib_gather:
            r.gather_wait()
        elif gatherer is not None:
            with torch.cuda.stream(lib_stream):
                finish_gather()
        r.wait()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    s0 = r.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                                          # the last batch's image is complete inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    s1 = r.stats()

    def delta(key):
        return s1[key] - s0[key]

    # (untimed) the image the timed loop produced — every batch since the reset above, warm-up included — for the parity check
    bench_image, bench_image_spp = None, 0
    if rank == 0 and not args.no_parity_check and (gatherer is None):
        bench_image, bench_image_spp = r.read_gathered() if use_lib_gather else r.read_accum()
        bench_image = bench_image.copy()

    # The reference's own loop shape (src/trace.rs:182-204): render sync_rate samples, read the accumulators back to
    # the host, repeat.  Timed separately — it is not `value` (inputs and outputs of `value` stay in HBM) — so that the
    # PCIe-inclusive rate is a measurement too.  One device-side un-tile + one DMA into pinned memory per read-back.
    readback = None
    if world_size == 1 and not args.no_readback:
        host_image = np.empty((H, W, 4), np.float32)
        r.reset(seeds)
        r.render(args.spp_per_step)
        r.read_accum(host_image)
        sa = r.stats()
        ta = time.perf_counter()
        for _ in range(args.steps):
            r.render(args.spp_per_step)
            r.read_accum(host_image)
        tb = time.perf_counter() - ta
        sb = r.stats()
        rb_rays = (sb["extension_rays"] - sa["extension_rays"]) + (sb["shadow_rays_traced"] - sa["shadow_rays_traced"])
        readback = {"loop": "rpt_render(spp_per_step) -> rpt_read_accum (host buffer), as src/trace.rs:182-204",
                    "ms_per_step": round(tb / args.steps * 1e3, 4), "value": round(rb_rays / tb / 1e6, 3), "unit": "Mrays/s",
                    "all_samples_arrived": bool((host_image[..., 3] == float(args.spp_per_step * (args.steps + 1))).all())}
        # the same loop with the read-back of batch k overlapped with the rendering of batch k+1 (rpt_comm_init_local: snapshot,
        # second stream, un-tile, DMA into pinned memory; the host sees every batch, one batch later)
        if not args.with_gather:
            r.comm_init_local()
            r.reset(seeds)
            r.render_async(args.spp_per_step)
            r.gather_async()
            r.gather_wait()                                  # batch 0 rendered and snapshotted before the clock starts
            sa = r.stats()
            ta = time.perf_counter()
            for _ in range(args.steps):                      # timed: `steps` batches rendered, `steps` + 1 images read
                r.render_async(args.spp_per_step)
                r.read_gathered(host_image)
                r.gather_async()
            _, seen = r.read_gathered(host_image)            # the last batch: nothing left to hide it behind
            tb = time.perf_counter() - ta
            sb = r.stats()
            rb_rays = (sb["extension_rays"] - sa["extension_rays"]) + (sb["shadow_rays_traced"] - sa["shadow_rays_traced"])
            readback["overlapped"] = {
                "loop": "rpt_render_async(k+1) ; rpt_read_gathered(k) ; rpt_gather_async: batch k reaches the host while k+1 renders",
                "ms_per_step": round(tb / args.steps * 1e3, 4), "value": round(rb_rays / tb / 1e6, 3), "unit": "Mrays/s",
                "all_samples_arrived": bool(seen == args.spp_per_step * (args.steps + 1) and (host_image[..., 3] == float(seen)).all())}

    local = torch.tensor([elapsed, float(delta("extension_rays")), float(delta("shadow_rays_traced")), float(delta("samples")),
                          float(delta("sky_evals")), float(delta("shadow_rays_elided"))], dtype=torch.float64, device=comm_device if world_size > 1 else device)
    if world_size > 1:
        tmax = local[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = local[1:].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed_max = float(tmax[0])
        n_ext, n_shadow, n_samples, n_sky, n_elided = (float(x) for x in sums)
    else:
        elapsed_max = elapsed
        n_ext, n_shadow, n_samples, n_sky, n_elided = (float(x) for x in local[1:])

    if rank != 0:
        if world_size > 1:
            dist.destroy_process_group()
        return

    rays = n_ext + n_shadow
    mrays = rays / elapsed_max / 1e6
    n_mis = (n_shadow + n_elided) if cfg.nee == 1 else 0.0                   # the MIS carry is written per NEE evaluation, walked or not

    # --- roofline of the dominant kernel, from the HIP events recorded in this run (rank 0's stream)
    cus, clock_mhz = hip.device_info(local_rank)
    pipeline_bytes = algorithmic_bytes(n_ext, n_shadow, n_mis, n_samples)
    pipeline_gbs = pipeline_bytes / elapsed_max / 1e9
    share = r.local_pixels() / float(W * H)                            # rank 0's part of the image (its kernels are the ones timed)
    # the kept PMC passes are whole-image launches of 32-sample batches on one GPU: a launch of this run covers share x spp_per_step / 32 as many slots
    roofline, whole = stage_roofline(hip, args.workload, s0, s1, args.steps, elapsed_max, cus, clock_mhz, pipeline_bytes * share, share * args.spp_per_step / 32.0)

    # --- CPU baseline: the oracle (a port of trace_cpu) on this host's cores, bounded sample of the same workload
    cpu = None
    if not args.no_cpu_baseline and world_size == 1:                 # (rank 0 at N = 1 only: the scaling runs do not repeat the ~12 s CPU leg)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_ffi import Oracle
        orc = Oracle("rpt_math")
        osc = orc.scene(world)
        cores = usable_cores()
        rows = max(8, H // 16)
        rect = (0, (H - rows) // 2, W, (H - rows) // 2 + rows)
        _, _, st = orc.trace_cpu(cfg, osc, seeds, 1, rect=rect, threads=cores)     # calibration pass
        rate = (st.extension_rays + st.shadow_rays) / max(st.seconds, 1e-9)
        spp = 1
        full_rays_per_spp = (st.extension_rays + st.shadow_rays) * (H / rows)
        spp = int(max(1, min(64, args.cpu_seconds * rate / max(full_rays_per_spp, 1))))
        _, _, st = orc.trace_cpu(cfg, osc, seeds, spp, threads=cores)
        cpu_rays = st.extension_rays + st.shadow_rays
        cpu = {"value": round(cpu_rays / st.seconds / 1e6, 3), "unit": "Mrays/s", "cores": int(st.threads), "kind": "port",
               "sample": f"{scene}.glb {W}x{H} {spp} spp, same config and seeds, {st.seconds:.1f} s on {st.threads} threads",
               "samples_per_s": round(st.samples / st.seconds, 1)}

    # --- parity inside the benchmark run: windows of the image the timed loop rendered vs the CPU oracle at the same sample
    # count; a mismatch fails the bench.
    order, last_order = shadow_order_label(r), last_bounce_label(r)
    parity = None
    if bench_image is not None:
        parity = parity_windows(world, cfg, seeds, bench_image, bench_image_spp, args.spp_per_step * (args.warmup + args.steps), W, H)

    # --- the other single-GPU BASELINE workloads (C2 with MIS, C3, C4 as one GPU sees it, the C5 stand-in): after the headline, never
    # inside its timed region, each on a fresh context
    workloads = None
    if world_size == 1 and args.workload == "darkcornell" and not args.no_extra_workloads and not args.with_gather:
        r.close()
        workloads = {}
        for name in [w for w in args.extra_workloads.split(",") if w]:
            if name not in WORKLOADS:
                raise SystemExit(f"--extra-workloads: unknown workload {name}")
            t_w = time.perf_counter()
            workloads[name] = measure_single_gpu_workload(rpt, hip, name, args.extra_steps, args.extra_warmup, args.spp_per_step,
                                                          local_rank, cus, clock_mhz, parity=not args.no_parity_check,
                                                          measure_startup=WORKLOADS[name][0].startswith("procedural:"))
            workloads[name]["wall_s"] = round(time.perf_counter() - t_w, 1)

    out = {}
    if args.rehearsal:
        out["rehearsal"] = "NOT A MEASUREMENT: all ranks share one GPU, gather over tests/fake_rccl (shared memory), torch.distributed over gloo"
    out.update({
        "metric": "Mrays/s", "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": data_label(scene) + ("" if ":" in scene else " + blue-noise seeds; no synthetic geometry"),
        "config": {"workload": workload_label(scene, W, H, args.steps, args.spp_per_step, total_spp, cfg),
                   "spp_per_step": args.spp_per_step, "tiles": "64x64 round-robin" if world_size > 1 else "single GPU",
                   "kernel_sources": hip.build_fingerprint(), "driver": "ranks: one process per GPU (rpt_comm_init = ncclCommInitRank)" if world_size > 1 else "single GPU",
                   "gather": gather_impl, "collective_library": hip.comm_library() or None,
                   "rpt_comm_world": comm_world_seen, "shadow_order": order, "last_bounce_order": last_order},
        "samples_per_s": round(n_samples / elapsed_max, 1),
        "value_counts": ("extension rays + shadow rays WALKED on the device (since round 5; rounds 1-4 counted one shadow ray per NEE evaluation, walked or not: "
                         "compare across rounds with value_as_the_reference_counts, samples_per_s or ms_per_step — identical to `value` at nee = 0)"),
        "value_as_the_reference_counts": round((rays + n_elided) / elapsed_max / 1e6, 3),
        "rays": {"extension": int(n_ext), "shadow": int(n_shadow), "shadow_elided": int(n_elided), "sky_evals": int(n_sky),
                 "per_sample": round(rays / max(n_samples, 1), 4),
                 "counted": "rays walked on the device; shadow_elided = NEE evaluations whose shadow ray decides nothing (zero term whatever the walk finds) and is not walked"},
        "roofline": roofline,
        "pipeline_roofline": {"bound": "hbm", "achieved": round(pipeline_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(pipeline_gbs / HBM_PEAK_GBS, 6),
                              "formula": "128*N_ext + 96*N_shadow + 128*N_mis + 40*samples (SURVEY.md 8d)"},
        "cpu_baseline": cpu,
        "parity_check": parity,
        "readback": readback,
    })
    if whole:                       # (rank 0's kernels; at N > 1 the per-rank share of the algorithmic bytes is what they are compared with)
        out["pipeline_roofline"].update({k: whole[k] for k in ("traffic", "traffic_over_algorithmic", "traffic_is", "stages_without_counters")})
        if roofline and "valu" in roofline:
            roofline["valu"].update({"issue_frac": whole["valu"]["issue_frac"], "issue_frac_is": whole["valu"]["issue_frac_is"],
                                     "batch_wave_instructions": whole["valu"]["wave_instructions_per_batch"],
                                     "batch_simd_cycles_per_wave_instruction": whole["valu"]["simd_cycles_per_wave_instruction"]})
    if workloads is not None:
        out["workloads"] = workloads
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world_size > 1:
        dist.destroy_process_group()
    failed = [] if parity_ok(parity) else [(args.workload, parity)]
    failed += [(n, w["parity_check"]) for n, w in (workloads or {}).items() if not parity_ok(w["parity_check"])]
    for n, pc in failed:
        print(f"bench: PARITY CHECK FAILED ({n}): " + json.dumps(pc), file=sys.stderr)
    if failed:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
