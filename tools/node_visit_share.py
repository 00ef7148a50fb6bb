#!/usr/bin/env python3
"""Which part of a mid-size BVH would a top-of-tree LDS image have to hold?  (CPU analysis, no GPU.)

gfx950 has 160 KB of LDS per CU; the walks of scenes too large for the 32 KB image (VeachMIS: 2 891 child pairs, PBRTest: 23 818) read
every node through L1 / L2 and wait two thirds of their cycles.  A 1 024-thread workgroup per CU could hold ~110 KB of pair records:
~1 100 pairs in today's sign-selected plane-record format (100 B per pair), ~1 950 as compact boxes (56 B per pair).
This counts, with the oracle's walk (oracle_node_histogram: node pops of the nearest-hit and the any-hit walks of real paths, the
BASELINE cameras), the share of INNER-node visits (= child-pair tests) that fall into the first K pairs of three static orders:
  bfs    breadth-first from the root
  area   pairs sorted by the surface area of their parent box (the SAH's visit probability; known at upload, camera independent)
  best   pairs sorted by the measured visit count itself (the ceiling for any static choice, camera DEPENDENT)
usage: python tools/node_visit_share.py [scene ...]
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_ffi import Oracle  # noqa: E402

rpt = importlib.import_module("rust-path-tracer_amd")
CASES = {"VeachMIS": (192, 108, 4, {"nee": 1}), "PBRTest": (128, 128, 4, {}), "FurnaceTest": (128, 128, 4, {"nee": 1}), "DarkCornell": (96, 96, 4, {"nee": 1})}
KS = (127, 256, 512, 1024, 1100, 1700, 1950, 4096)


def main():
    orc = Oracle()
    for scene in (sys.argv[1:] or ["VeachMIS", "PBRTest"]):
        W, H, spp, over = CASES[scene]
        w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
        sc = orc.scene(w)
        cfg = rpt.default_config(W, H, **over)
        seeds = rpt.blue_noise_seeds(W, H)
        nodes_f = w.nodes.view(np.float32).reshape(-1, 8)
        nodes_u = w.nodes.view(np.uint32).reshape(-1, 8)
        n = len(nodes_u)
        hn, ha = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        rc = orc.lib.oracle_node_histogram(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(spp),
                                           hn.ctypes.data_as(C.c_void_p), ha.ctypes.data_as(C.c_void_p))
        assert rc == 0
        inner = nodes_u[:, 3] == 0
        pairs = np.flatnonzero(inner)                    # a pair is identified by its parent (the node whose pop tests it)
        depth = np.zeros(n, np.int64)
        order_bfs, queue = [], [0]
        while queue:
            nxt = []
            for p in queue:
                if inner[p]:
                    order_bfs.append(p)
                    l = int(nodes_u[p, 7])
                    depth[l] = depth[l + 1] = depth[p] + 1
                    nxt += [l, l + 1]
            queue = nxt
        order_bfs = np.array(order_bfs)
        ext = nodes_f[:, 4:7] - nodes_f[:, 0:3]
        area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
        order_area = pairs[np.argsort(-area[pairs], kind="stable")]
        print(f"{scene} {W}x{H} x {spp} spp, nee={cfg.nee}: {n} nodes, {len(pairs)} child pairs, depth {depth.max()}; "
              f"nearest-hit walks: {int(hn[inner].sum())} pair tests + {int(hn[~inner].sum())} leaf visits; any-hit walks: {int(ha[inner].sum())} + {int(ha[~inner].sum())}")
        for name, h in (("nearest", hn), ("any-hit", ha)):
            tot = float(h[inner].sum())
            if tot == 0:
                continue
            order_best = pairs[np.argsort(-h[pairs].astype(np.int64), kind="stable")]
            for oname, order in (("bfs", order_bfs), ("area", order_area), ("best", order_best)):
                c = np.cumsum(h[order].astype(np.float64)) / tot
                print(f"    {name:8s} {oname:5s}: " + "  ".join(f"K={k}: {c[min(k, len(c)) - 1]:.1%}" for k in KS if k <= len(c) * 2))
            # a walk that leaves the image: how many of a ray's pair tests come AFTER its first test outside the top K (area order)?
        # the subtree closure: an image is only useful if a pair's parent is in it too (the walk enters from the root)
        for k in (1100, 1950):
            top = set(order_area[:k].tolist())
            closed = sum(1 for p in order_area[:k] if p == 0 or _parent(nodes_u, inner)[p] in top)
            print(f"    area order, K = {k}: {closed} of {min(k, len(order_area))} pairs have their parent in the set")


_parent_cache = {}


def _parent(nodes_u, inner):
    key = id(nodes_u)
    if key not in _parent_cache:
        par = np.full(len(nodes_u), -1, np.int64)
        for p in np.flatnonzero(inner):
            l = int(nodes_u[p, 7])
            par[l] = par[l + 1] = p
        _parent_cache[key] = par
    return _parent_cache[key]


if __name__ == "__main__":
    main()
