#!/usr/bin/env python3
"""How many inner-node steps of a wave of PRIMARY rays are wave-uniform?  (CPU analysis, no GPU.)

The streamed global-memory walks are bound by the CU's texture-address unit.  A wave of the first iteration is an 8 x 8 pixel block
at one sample index: its 64 camera rays are nearly parallel, so near the top of the tree every lane stands on the SAME node — a visit that one
scalar load (s_load_dwordx16 of the 64-byte pair record, scalar cache, no TA cycles) could serve instead of 4 vector loads.  This replays the
walks of random 8 x 8 blocks of the BASELINE images at their real resolution with the kernel's one-body-per-trip rule and counts the inner trips
in which all participating lanes stand on one node.  usage: python tools/uniform_visit_share.py [scene ...]
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
CASES = {"PBRTest": (2048, 2048, {}), "VeachMIS": (1920, 1080, {"nee": 1}), "FurnaceTest": (256, 256, {})}


def main():
    orc = Oracle()
    rng = np.random.default_rng(7)
    for scene in (sys.argv[1:] or ["PBRTest", "VeachMIS"]):
        W, H, over = CASES[scene]
        w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
        sc = orc.scene(w)
        cfg = rpt.default_config(W, H, **over)
        seeds = rpt.blue_noise_seeds(W, H)
        nodes_u = w.nodes.view(np.uint32).reshape(-1, 8)
        leaf = nodes_u[:, 3] > 0
        for bounce in (0, 1):
            rays = np.zeros((W * H, 6), np.float32)
            valid = np.zeros(W * H, np.uint8)
            orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                                     rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
            tot_inner = tot_uniform = tot_lanes = tot_uniform_lanes = 0
            blocks = 0
            for _ in range(400):
                bx, by = int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))
                idx = np.array([(by * 8 + y) * W + bx * 8 + x for y in range(8) for x in range(8)])
                v = valid[idx] == 1
                if v.sum() < 8:
                    continue
                o = np.ascontiguousarray(rays[idx, :3]); d = np.ascontiguousarray(rays[idx, 3:])
                cap = 256
                nl = np.zeros((64, cap), np.uint32); ln = np.zeros(64, np.uint32)
                orc.lib.oracle_trace_nodes(C.byref(sc), C.c_size_t(64), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), None,
                                           nl.ctypes.data_as(C.c_void_p), C.c_uint32(cap), ln.ctypes.data_as(C.c_void_p))
                ln = np.where(v, np.minimum(ln, cap), 0)
                pos = np.zeros(64, np.int64)
                while True:
                    act = pos < ln
                    if not act.any():
                        break
                    cur = nl[np.arange(64), np.minimum(pos, cap - 1)]
                    at_leaf = act & leaf[cur]
                    at_inner = act & ~leaf[cur]
                    if at_leaf.sum() > at_inner.sum():
                        pos[at_leaf] += 1
                        continue
                    n = int(at_inner.sum())
                    uni = len(np.unique(cur[at_inner])) == 1
                    tot_inner += 1; tot_lanes += n
                    if uni:
                        tot_uniform += 1; tot_uniform_lanes += n
                    pos[at_inner] += 1
                blocks += 1
            if blocks:
                print(f"{scene} {W}x{H} bounce {bounce}: {blocks} blocks of 8 x 8 pixels; inner trips that are wave-uniform {tot_uniform / max(tot_inner, 1):.1%} "
                      f"(of the lane-visits {tot_uniform_lanes / max(tot_lanes, 1):.1%}); lanes per inner trip {tot_lanes / max(tot_inner, 1):.1f}")


if __name__ == "__main__":
    main()
