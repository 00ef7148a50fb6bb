#!/usr/bin/env python3
"""Replay of the streamed any-hit (shadow) walks under the visiting orders the reference's `.hit` leaves free — see tools/anyhit_order_sim.cpp.

usage: python tools/anyhit_order_sim.py [scene ...] [--blocks N] [--samples S] [--bounces 0,1,2,3]
Scenes: DarkCornell (LDS walk: trips 16 / refill 16), VeachMIS, scatter (global-memory walks: trips 8 / refill 24).
Rays: the shadow rays the oracle's trace_pixel traces for S samples of runs of 8 x 8 pixel blocks, in the order the shade stage queues them
(a wave's 64 slots = one block at one sample index; sample-major inside a block run), cut into spans of 512 queue positions = one wave's share.
"""
import argparse
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_ffi import Oracle, _p  # noqa: E402

rpt = importlib.import_module("rust-path-tracer_amd")

SCENES = {
    # name: (width, height, config overrides, trips, refill, kind)
    "DarkCornell": (1024, 1024, {"nee": 1}, 16, 16, "lds"),
    "VeachMIS": (1920, 1080, {"nee": 1}, 8, 24, "global"),
    "scatter": (2048, 2048, {"nee": 1, "cam_position": (0.0, 1.8, -0.9, 0.0)}, 8, 24, "global"),
    "deepbvh": (2048, 2048, {"nee": 1, "cam_position": (0.0, 2.5, -0.5, 0.0)}, 8, 24, "global"),
    "FurnaceTest": (256, 256, {"nee": 1}, 8, 24, "global"),
}
ORDERS = ["near (today)", "left first", "larger box first", "wave vote", "packet", "packet + refill", "threaded",
          "more opaque first", "smaller subtree", "opacity / nodes", "opaque area/nodes", "more tri area",
          "in range, near", "in range, opaque", "crossed, near", "crossed, opaque", "origin last, near", "origin last, opaq", "range,origin,near", "range,origin,opaq",
          "opaque x1.25|near", "opaque x2 | near", "opaque x4 | near", "opaque x8 | near",
          "learned all", "learned all/n^.5", "learned all/n", "learned first", "learned first/n^.5", "learned first/n"]
ORDER_CODE = [0, 1, 20, 3, 4, 5, 6, 21, 22, 23, 24, 25, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 50, 51, 52, 53, 54, 55]
DYNAMIC = range(12, 20)
FIELDS = ["rays", "occluded", "inner_trips", "inner_uniform", "inner_lanes", "leaf_trips", "leaf_iters", "leaf_lanes", "pop_trips", "refills", "skipped",
          "box_tests", "tri_tests", "max_stack", "visits_occluded", "visits_clear", "rays_clear"]

# VALU + SALU wave-instructions per body, read off the disassembly of the shipped kernels (tools/anyhit_body_costs.md has the listing ranges):
#   global: k_traverse_shadow_gstream<24,16,false>   lds: k_traverse_shadow_stream<16,1024>
COSTS = {
    #            inner body (vector loads), inner body (scalar loads), what a fixed order saves, per triangle iteration, per leaf trip, per trip, per refill pass, vmem per inner trip
    "global": {"inner": 118, "inner_uniform": 110, "order_saving": 6, "tri": 58, "leaf": 14, "trip": 14, "refill": 60, "vmem_inner": 4, "vmem_tri": 3, "vmem_pop": 1,
               "packet_inner": 96, "packet_tri": 56, "packet_leaf": 16, "packet_trip": 18, "thread_box": 62, "vmem_box": 2},
    "lds": {"inner": 98, "inner_uniform": 98, "order_saving": 5, "tri": 52, "leaf": 12, "trip": 12, "refill": 70, "vmem_inner": 0, "vmem_tri": 0, "vmem_pop": 0,
            "packet_inner": 90, "packet_tri": 52, "packet_leaf": 14, "packet_trip": 16, "thread_box": 56, "vmem_box": 0},
}


def build():
    out = os.path.join(ROOT, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libanyhit_sim.so")
    src = os.path.join(ROOT, "tools", "anyhit_order_sim.cpp")
    deps = [src, os.path.join(ROOT, "oracle", "rpt_oracle.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-std=c++20", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-msse4.1", "-pthread", "-shared", "-o", so, src], check=True)
    return C.CDLL(so)


def load_world(name):
    if name == "scatter":
        from scenes import scatter_scene
        return scatter_scene(1_000_000)
    if name == "deepbvh":
        from scenes import deep_bvh_scene
        return deep_bvh_scene(1_000_000)
    return rpt.World.from_path(rpt.fixture(name + ".glb"))


def block_pixels(bx, by):
    return [(by * 8 + y) << 16 | (bx * 8 + x) for y in range(8) for x in range(8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["DarkCornell", "VeachMIS"])
    ap.add_argument("--runs", type=int, default=60, help="runs of 4 horizontally adjacent 8 x 8 blocks")
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--bounces", default="0,1,2,3")
    ap.add_argument("--span", type=int, default=512)
    ap.add_argument("--all-rays", action="store_true", help="include the shadow rays whose NEE term is zero whatever the walk finds (the device elides them, k_shade.h)")
    args = ap.parse_args()
    sim = build()
    orc = Oracle()
    for name in args.scenes:
        W, H, over, trips, refill, kind = SCENES[name]
        cost = COSTS[kind]
        world = load_world(name)
        sc = orc.scene(world)
        cfg = rpt.default_config(W, H, **over)
        seeds = rpt.blue_noise_seeds(W, H)
        rng = np.random.default_rng(5)
        runs = [(int(rng.integers(0, W // 8 - 4)), int(rng.integers(0, H // 8))) for _ in range(args.runs)]
        print(f"\n=== {name} {W}x{H} nee = MIS: {args.runs} runs of 4 blocks x {args.samples} samples, spans of {args.span}; walk = {kind} (trips {trips}, refill at {refill} idle lanes)")
        total = {b: np.zeros((len(ORDERS), len(FIELDS)), np.float64) for b in ["all"]}
        per_bounce = {}
        spans_of = {}
        for bounce in [int(b) for b in args.bounces.split(",")]:
            spans_of[bounce] = []
            for (bx, by) in runs:
                stream = []
                for k in range(4):
                    pix = np.array(block_pixels(bx + k, by), np.uint32)
                    for s in range(args.samples):
                        rays = np.zeros((64, 8), np.float32)
                        valid = np.zeros(64, np.uint8)
                        sim.sim_dump_shadow_rays(C.byref(cfg), C.byref(sc), _p(seeds), C.c_uint32(s), C.c_uint32(bounce), _p(pix), C.c_size_t(64), _p(rays), _p(valid))
                        stream.append(rays[(valid == 1) | ((valid == 2) & args.all_rays)])
                stream = np.concatenate(stream)
                for at in range(0, len(stream), args.span):
                    span = np.ascontiguousarray(stream[at:at + args.span])
                    if len(span) >= 64:
                        spans_of[bounce].append(span)
        # the learned orders train on every second span of every bounce (what a first batch would see) and are evaluated on all of them
        train = [sp for b in spans_of for sp in spans_of[b][::2]]
        for i, sp in enumerate(train):
            sim.sim_learn(C.byref(sc), _p(sp), C.c_uint32(len(sp)), 1 if i == len(train) - 1 else 0)
        for bounce, spans in spans_of.items():
            acc = np.zeros((len(ORDERS), len(FIELDS)), np.float64)
            for span in spans:
                ref_hit = None
                for o in range(len(ORDERS)):
                    out = np.zeros(24, np.uint64)
                    hit = np.zeros(len(span), np.uint8)
                    nf = sim.sim_wave(C.byref(sc), _p(span), C.c_uint32(len(span)), ORDER_CODE[o], trips, refill, _p(out), _p(hit))
                    assert nf == len(FIELDS), nf
                    if ref_hit is None:
                        ref_hit = hit
                    assert np.array_equal(hit, ref_hit), (name, bounce, ORDERS[o])        # .hit is order-independent
                    row = out[:nf].astype(np.float64)
                    row[FIELDS.index("max_stack")] = 0
                    acc[o] += row
                    acc[o, FIELDS.index("max_stack")] = max(acc[o, FIELDS.index("max_stack")], float(out[FIELDS.index("max_stack")]))
            ms = max(total["all"][:, FIELDS.index("max_stack")].max(), acc[:, FIELDS.index("max_stack")].max())
            total["all"] += acc
            total["all"][:, FIELDS.index("max_stack")] = ms
            report(f"bounce {bounce} ({len(spans)} spans)", acc, cost)
        report("all bounces", total["all"], cost)


def report(title, acc, cost):
    f = {k: i for i, k in enumerate(FIELDS)}
    rays = acc[0, f["rays"]]
    if rays == 0:
        print(f"  {title}: no shadow rays")
        return
    print(f"  {title}: {int(rays)} rays, {100 * acc[0, f['occluded']] / rays:.1f} % occluded")
    print(f"    {'order':18s} {'inner trips':>11s} {'uniform':>8s} {'lanes/in':>8s} {'leaf trips':>10s} {'tri iters':>9s} {'lanes/lf':>8s} {'visits occ':>10s} {'visits clr':>10s} "
          f"{'wave-inst/ray':>13s} {'vs today':>8s} {'vmem/ray':>8s} {'vs today':>8s} {'stack':>5s}")
    base_inst = base_vmem = None
    for o, oname in enumerate(ORDERS):
        a = acc[o]
        it, iu, il = a[f["inner_trips"]], a[f["inner_uniform"]], a[f["inner_lanes"]]
        lt, li, ll = a[f["leaf_trips"]], a[f["leaf_iters"]], a[f["leaf_lanes"]]
        trips = it + lt
        if oname.startswith("packet"):
            inst = it * cost["packet_inner"] + li * cost["packet_tri"] + lt * cost["packet_leaf"] + (trips + a[f["skipped"]]) * cost["packet_trip"] + a[f["refills"]] * cost["refill"]
            vmem = 0.0
            lds_bytes = 16 * a[f["max_stack"]]
        elif oname == "threaded":
            inst = it * cost["thread_box"] + li * cost["tri"] + lt * cost["leaf"] + trips * cost["trip"] + a[f["refills"]] * cost["refill"]
            vmem = it * cost["vmem_box"] + li * cost["vmem_tri"]
        else:
            fixed = o in (1, 2) or 7 <= o < 12 or o >= 24
            body = cost["inner"] - (cost["order_saving"] if fixed else 0)
            body_u = cost["inner_uniform"] - (cost["order_saving"] if fixed else 0)
            inst = (it - iu) * body + iu * body_u + li * cost["tri"] + lt * cost["leaf"] + trips * cost["trip"] + a[f["refills"]] * cost["refill"]
            vmem = (it - iu) * cost["vmem_inner"] + li * cost["vmem_tri"] + a[f["pop_trips"]] * cost["vmem_pop"]
        inst /= rays
        vmem /= rays
        if base_inst is None:
            base_inst, base_vmem = inst, vmem
        vo = a[f["visits_occluded"]] / max(a[f["occluded"]], 1)
        vc = a[f["visits_clear"]] / max(a[f["rays_clear"]], 1)
        print(f"    {oname:18s} {it / rays:11.2f} {100 * iu / max(it, 1):7.1f}% {il / max(it, 1):8.1f} {lt / rays:10.2f} {li / rays:9.2f} {ll / max(lt, 1):8.1f} {vo:10.1f} {vc:10.1f} "
              f"{inst:13.1f} {100 * (inst / base_inst - 1):+7.1f}% {vmem:8.2f} {100 * (vmem / base_vmem - 1) if base_vmem else 0:+7.1f}% {int(a[f['max_stack']]):5d}")


if __name__ == "__main__":
    main()
