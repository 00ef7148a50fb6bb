#!/usr/bin/env python3
"""What would binning the packed shade stage by path kind buy?  (CPU analysis, no GPU.)

k_shade<NEE, TEXTURED, COMPACT = true> (k_shade.h): a workgroup owns 2 048 consecutive slots — with 32 samples in flight that is ONE 8 x 8 pixel block at its
32 sample indices — packs the slots traversed in this iteration into an LDS list in slot order and shades the list 256 at a time, 4 waves of 64.  Within a
wave a miss (park the slot, queue it for the sky stage), a path that ends on an emitter, a diffuse lobe (+ the NEE set-up) and a specular lobe share the ~1 400
instruction body: PMC lane utilisation 45 % on VeachMIS / PBRTest (profiles/r04_*_pmc_sq.txt).

This replays the stage from the oracle's record of what every bounce of every sample IS (oracle_path_kinds: kernels/src/lib.rs:64-181 — the `break` sites are
the kinds) and prices a wave as the sum of the bodies at least one of its lanes needs, in VALU instructions read off the disassembly:
  today      the list in slot order
  miss|hit   misses packed from the back of the list, everything else from the front (two-sided fill: no second list)
  3 bins     misses | emitter hits | surfaces   (needs the emitter bit of the hit triangle at packing time: one byte per triangle)
  4 bins     misses | emitter hits | diffuse | specular   (the ideal the review asks about: the lobe is only known ~350 instructions into the surface body)
usage: python tools/shade_bin_sim.py [scene ...]
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_ffi import Oracle, _p  # noqa: E402

rpt = importlib.import_module("rust-path-tracer_amd")

SCENES = {"VeachMIS": (1920, 1080, {"nee": 1}), "PBRTest": (2048, 2048, {}), "DarkCornell": (1024, 1024, {"nee": 1})}
MISS, EMIT, DIFF, SPEC = 1, 2, 3, 4
# VALU instructions of each part of shade_slot (k_shade<1, false, true> / <0, false, true>, gfx950 disassembly, rounded): what a wave issues when at least one
# of its lanes needs the part
COST = {
    1: {"fixed": 40, "miss": 25, "hit_pre": 90, "emit": 110, "surface_pre": 330, "diffuse": 300, "specular": 150, "surface_post": 240, "nee": 430, "tail": 80},
    0: {"fixed": 40, "miss": 25, "hit_pre": 90, "emit": 30, "surface_pre": 330, "diffuse": 300, "specular": 150, "surface_post": 240, "nee": 0, "tail": 70},
}


def wave_cost(kinds, cost):
    """(instructions issued, lane-instructions used) of one wave holding these kinds"""
    n = {k: int((kinds == k).sum()) for k in (MISS, EMIT, DIFF, SPEC)}
    hits = n[EMIT] + n[DIFF] + n[SPEC]
    surf = n[DIFF] + n[SPEC]
    live = n[MISS] + hits
    issued = used = 0
    for part, lanes in (("fixed", live), ("miss", n[MISS]), ("hit_pre", hits), ("emit", n[EMIT]), ("surface_pre", surf), ("diffuse", n[DIFF]), ("specular", n[SPEC]),
                        ("surface_post", surf), ("nee", n[DIFF]), ("tail", surf)):
        if lanes > 0 and cost[part] > 0:
            issued += cost[part]
            used += cost[part] * lanes
    return issued, used


def replay(lists, cost):
    issued = used = waves = 0
    for lst in lists:                       # one list = what one workgroup shades in one iteration, already ordered
        for at in range(0, len(lst), 64):
            i, u = wave_cost(lst[at:at + 64], cost)
            issued += i; used += u; waves += 1
    return issued, used, waves


def orders(k):
    """the packed list of one workgroup (kinds in slot order) under each scheme; chunks of 256 keep their 4-wave structure"""
    out = {"today": k}
    out["miss|hit"] = np.concatenate([k[k != MISS], k[k == MISS]])
    out["3 bins"] = np.concatenate([k[(k == DIFF) | (k == SPEC)], k[k == EMIT], k[k == MISS]])
    out["4 bins"] = np.concatenate([k[k == DIFF], k[k == SPEC], k[k == EMIT], k[k == MISS]])
    return out


def main():
    scenes = sys.argv[1:] or ["VeachMIS", "PBRTest"]
    orc = Oracle()
    for name in scenes:
        W, H, over = SCENES[name]
        cfg = rpt.default_config(W, H, **over)
        cost = COST[1 if cfg.nee else 0]
        world = rpt.World.from_path(rpt.fixture(name + ".glb"))
        sc = orc.scene(world)
        seeds = rpt.blue_noise_seeds(W, H)
        rng = np.random.default_rng(3)
        S = 32
        blocks = [(int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))) for _ in range(48)]
        per_bounce = {}
        for (bx, by) in blocks:
            pix = np.array([(by * 8 + y) << 16 | (bx * 8 + x) for y in range(8) for x in range(8)], np.uint32)
            kinds = np.zeros((S, 64, 8), np.uint8)
            for s in range(S):
                orc.lib.oracle_path_kinds(C.byref(cfg), C.byref(sc), _p(seeds), C.c_uint32(s), _p(pix), C.c_size_t(64), _p(kinds[s]))
            for b in range(cfg.max_bounces):
                k = kinds[:, :, b].reshape(-1)          # slot order inside the chunk: sample-major
                k = k[k != 0]
                if len(k):
                    per_bounce.setdefault(b, []).append(k)
        print(f"\n=== {name} {W}x{H} nee={cfg.nee}: {len(blocks)} workgroups (8 x 8 pixels x {S} samples), wave = 64 list entries")
        tot = {}
        for b, lists in sorted(per_bounce.items()):
            allk = np.concatenate(lists)
            share = {n: 100.0 * (allk == c).mean() for n, c in (("miss", MISS), ("emitter", EMIT), ("diffuse", DIFF), ("specular", SPEC))}
            print(f"  bounce {b}: {len(allk)} traversed slots: " + ", ".join(f"{n} {v:.1f} %" for n, v in share.items()))
            for scheme in ("today", "miss|hit", "3 bins", "4 bins"):
                i, u, w = replay([orders(k)[scheme] for k in lists], cost)
                t = tot.setdefault(scheme, [0, 0, 0])
                t[0] += i; t[1] += u; t[2] += w
                print(f"      {scheme:9s} lanes {100.0 * u / (64 * i):5.1f} %   {i / len(allk):7.1f} wave-instructions per slot")
        base = tot["today"][0]
        print("  all bounces:")
        for scheme, (i, u, w) in tot.items():
            print(f"      {scheme:9s} lanes {100.0 * u / (64 * i):5.1f} %   wave-instructions {100.0 * (i / base - 1):+6.1f} %")


if __name__ == "__main__":
    main()
