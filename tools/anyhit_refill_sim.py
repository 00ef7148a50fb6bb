#!/usr/bin/env python3
"""Round 6, review item 6 (C3: the shadow walk of VeachMIS at 46.6 % lanes): what would another REFILL cadence of the streamed any-hit walk buy, and what a
2-wide leaf test?  Replay (tools/anyhit_order_sim.cpp: the kernels' trip / majority / refill rules on the scene's real shadow rays in queue order, only the rays the
device walks) of the order the upload probe chose, under (trips between refill checks, idle lanes that trigger a refill) pairs around today's.

usage: python tools/anyhit_refill_sim.py [VeachMIS ...] [--runs 40]
"""
import argparse
import ctypes as C
import sys

import numpy as np

import anyhit_order_sim as A

VARIANTS = [(8, 24), (8, 16), (8, 8), (4, 24), (4, 16), (4, 8), (2, 8), (16, 24), (16, 32), (8, 32), (1, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["VeachMIS"])
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--span", type=int, default=512)
    args = ap.parse_args()
    sim = A.build()
    orc = A.Oracle()
    f = {k: i for i, k in enumerate(A.FIELDS)}
    for name in args.scenes:
        W, H, over, trips0, refill0, kind = A.SCENES[name]
        cost = A.COSTS[kind]
        world = A.load_world(name)
        sc = orc.scene(world)
        cfg = A.rpt.default_config(W, H, **over)
        seeds = A.rpt.blue_noise_seeds(W, H)
        rng = np.random.default_rng(5)
        runs = [(int(rng.integers(0, W // 8 - 4)), int(rng.integers(0, H // 8))) for _ in range(args.runs)]
        spans = []
        for bounce in (0, 1, 2, 3):
            for (bx, by) in runs:
                stream = []
                for k in range(4):
                    pix = np.array(A.block_pixels(bx + k, by), np.uint32)
                    for s in range(args.samples):
                        rays = np.zeros((64, 8), np.float32)
                        valid = np.zeros(64, np.uint8)
                        sim.sim_dump_shadow_rays(C.byref(cfg), C.byref(sc), A._p(seeds), C.c_uint32(s), C.c_uint32(bounce), A._p(pix), C.c_size_t(64), A._p(rays), A._p(valid))
                        stream.append(rays[valid == 1])
                stream = np.concatenate(stream)
                for at in range(0, len(stream), args.span):
                    span = np.ascontiguousarray(stream[at:at + args.span])
                    if len(span) >= 64:
                        spans.append(span)
        print(f"\n=== {name} {W}x{H} nee = MIS, all bounces: {len(spans)} spans of up to {args.span} rays (the rays the device walks); walk = {kind}; today: trips {trips0}, refill at {refill0} idle lanes")
        print(f"    {'order':20s} {'trips/refill':>12s} {'inner trips':>11s} {'lanes/in':>8s} {'leaf trips':>10s} {'tri iters':>9s} {'lanes/lf':>8s} {'refills/ray':>11s} {'wave-inst/ray':>13s} {'vs today':>8s}"
              f" {'2-wide leaves':>13s}")
        for oname, code in (("near first", 0), ("more opaque first", 21)):
            base = None
            for (trips, refill) in VARIANTS:
                acc = np.zeros(len(A.FIELDS), np.float64)
                for span in spans:
                    out = np.zeros(24, np.uint64)
                    hit = np.zeros(len(span), np.uint8)
                    sim.sim_wave(C.byref(sc), A._p(span), C.c_uint32(len(span)), code, trips, refill, A._p(out), A._p(hit))
                    acc += out[:len(A.FIELDS)].astype(np.float64)
                rays = acc[f["rays"]]
                it, iu, il = acc[f["inner_trips"]], acc[f["inner_uniform"]], acc[f["inner_lanes"]]
                lt, li, ll = acc[f["leaf_trips"]], acc[f["leaf_iters"]], acc[f["leaf_lanes"]]
                fixed = code != 0
                body = cost["inner"] - (cost["order_saving"] if fixed else 0)
                body_u = cost["inner_uniform"] - (cost["order_saving"] if fixed else 0)
                inst = ((it - iu) * body + iu * body_u + li * cost["tri"] + lt * cost["leaf"] + (it + lt) * cost["trip"] + acc[f["refills"]] * cost["refill"]) / rays
                # a leaf body that tests two triangles per iteration: half the iterations (rounded up per leaf trip: at least one per trip), each 1.85 x the
                # instructions (the ray set-up, the loop and the accept bookkeeping are shared; the two Moeller-Trumbore bodies are not)
                li2 = np.maximum(lt, (li + lt) / 2.0)
                inst2 = ((it - iu) * body + iu * body_u + li2 * cost["tri"] * 1.85 + lt * cost["leaf"] + (it + lt) * cost["trip"] + acc[f["refills"]] * cost["refill"]) / rays
                if base is None:
                    base = inst
                print(f"    {oname:20s} {f'{trips}/{refill}':>12s} {it / rays:11.2f} {il / max(it, 1):8.1f} {lt / rays:10.2f} {li / rays:9.2f} {ll / max(lt, 1):8.1f} {acc[f['refills']] / rays:11.3f} "
                      f"{inst:13.1f} {100 * (inst / base - 1):+7.1f}% {100 * (inst2 / inst - 1):+12.1f}%")


if __name__ == "__main__":
    sys.exit(main())
