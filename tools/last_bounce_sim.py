#!/usr/bin/env python3
"""What does the LAST extension ray of a path have to find when there is no NEE?  (CPU analysis, no GPU.)

kernels/src/lib.rs:62-109 at bounce max_bounces - 1 with nee = 0: a miss adds the sky (:66-79); a hit on the FRONT of a triangle whose material emits adds
throughput x emission (:86-101); every other hit adds nothing, and what the loop body computes after it (:112-181) is read by nobody.  So a ray that passes the
Moller-Trumbore test of NO emissive triangle only has to answer "hit or miss" — and the reference's own walk (intersection.rs:177-234) answers that at its FIRST
accepted triangle: result.hit never becomes false again.  This replays the rays of the last bounce through the oracle's walk twice, to the end and to the first
accept, and counts node visits of both, and the share of rays that pass the test of an emissive triangle (they keep the whole walk).

usage: python tools/last_bounce_sim.py [scene ...]      (DarkCornell PBRTest by default; 256 x 256 pixels of the BASELINE view, 4 samples)
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle_ffi import Oracle, _p  # noqa: E402

rpt = importlib.import_module("rust-path-tracer_amd")


def mt_pass(o, d, a, b, c):
    """numpy float32 Moller-Trumbore (intersection.rs:9-54): does ray pass?  (an estimate: the op order is the reference's, fused or not is numpy's)"""
    e1, e2 = b - a, c - a
    pv = np.cross(d, e2)
    det = (e1 * pv).sum(-1)
    ok = np.abs(det) >= 1e-6
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0).astype(np.float32)
    tv = o - a
    u = (tv * pv).sum(-1) * inv
    ok &= (u >= 0) & (u <= 1)
    qv = np.cross(tv, e1)
    v = (d * qv).sum(-1) * inv
    ok &= (v >= 0) & (u + v <= 1)
    t = (e2 * qv).sum(-1) * inv
    return ok & (t >= 0)


def main():
    scenes = sys.argv[1:] or ["DarkCornell", "PBRTest"]
    orc = Oracle()
    for name in scenes:
        W = H = 256
        cfg = rpt.default_config(W, H)
        world = rpt.World.from_path(rpt.fixture(name + ".glb"))
        sc = orc.scene(world)
        verts = np.ascontiguousarray(world.per_vertex["vertex"], np.float32).reshape(-1, 4)[:, :3]
        em = np.ascontiguousarray(world.materials["emissive"], np.float32)[:, :3]
        em_mat = [m for m in range(len(em)) if np.any(em[m] != 0) or np.any(np.isnan(em[m]))]      # `emissive.xyz() != Vec3::ZERO` (lib.rs:86)
        tri_mat = world.indices["material"]
        em_tris = np.nonzero(np.isin(tri_mat, em_mat))[0]
        print(f"\n=== {name}: {len(world.indices)} triangles, {len(em_tris)} of them emissive (materials {em_mat}); bounces {cfg.min_bounces}/{cfg.max_bounces}, nee {cfg.nee}")
        last = cfg.max_bounces - 1
        tot = {}
        for s in range(4):
            seeds = rpt.blue_noise_seeds(W, H)
            seeds = seeds.copy()
            seeds["n"] += s                        # (the sample index: kernels/src/rng.rs:20-63)
            for bounce in range(cfg.max_bounces):
                rays = np.zeros((W * H, 6), np.float32)
                valid = np.zeros(W * H, np.uint8)
                orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), _p(seeds), C.c_uint32(bounce), _p(rays), _p(valid))
                r = rays[valid == 1]
                n = len(r)
                if n == 0:
                    continue
                o, d = np.ascontiguousarray(r[:, :3]), np.ascontiguousarray(r[:, 3:])
                cap = 4096
                ev = np.zeros((n, cap), np.uint8); ln = np.zeros(n, np.uint32)
                orc.lib.oracle_trace_events(C.byref(sc), C.c_size_t(n), _p(o), _p(d), _p(ev), C.c_uint32(cap), _p(ln))
                full = int(ln.sum())
                nl = np.zeros((n, 1), np.uint32); l1 = np.zeros(n, np.uint32)
                far = np.full(n, 1e6, np.float32)
                orc.lib.oracle_trace_nodes(C.byref(sc), C.c_size_t(n), _p(o), _p(d), _p(far), _p(nl), C.c_uint32(1), _p(l1))
                may_emit = np.zeros(n, bool)
                for t in em_tris:
                    tri = world.indices[t]
                    a, b, c = (verts[int(tri[k])] for k in range(3))
                    may_emit |= mt_pass(o, d, a, b, c)
                mixed = int(np.where(may_emit, ln, l1).sum())
                k = tot.setdefault(bounce, [0, 0, 0, 0, 0])
                k[0] += n; k[1] += full; k[2] += int(l1.sum()); k[3] += mixed; k[4] += int(may_emit.sum())
        allfull = sum(v[1] for v in tot.values())
        for b, (n, full, first, mixed, me) in sorted(tot.items()):
            print(f"  bounce {b}: {n} rays, node visits per ray: whole walk {full / n:6.2f}   to the first accept {first / n:6.2f}   "
                  f"| rays passing an emissive triangle's test {100.0 * me / n:5.2f} %  -> mixed {mixed / n:6.2f} ({100.0 * (mixed / full - 1):+.1f} %)"
                  + ("   <- the last bounce" if b == last else ""))
        n, full, first, mixed, me = tot[last]
        print(f"  all bounces: node visits {allfull} -> {allfull - full + mixed} ({100.0 * ((allfull - full + mixed) / allfull - 1):+.1f} %) with the last bounce cut at its first accept")
        replay_orders(orc, sc, cfg, name, W, H, last)


def replay_orders(orc, sc, cfg, name, W, H, bounce):
    """The part of the walk up to the first accept is an any-hit walk (result.t = 1e6 throughout): free in its visiting order like a shadow query.  The rays of
    the last bounce through the streamed LDS kernel's replay (tools/anyhit_order_sim.cpp: waves of 64 lanes over spans of 512 rays, 16 trips between refills)
    under the reference's near-first order and the static orders csrc/shadow_order.h choose_last_order picks from."""
    import anyhit_order_sim as A
    sim = A.build()
    seeds = rpt.blue_noise_seeds(W, H)
    rays = np.zeros((W * H, 6), np.float32)
    valid = np.zeros(W * H, np.uint8)
    orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), _p(seeds), C.c_uint32(bounce), _p(rays), _p(valid))
    idx = np.array([(by * 8 + y) * W + bx * 8 + x for by in range(H // 8) for bx in range(W // 8) for y in range(8) for x in range(8)])   # a wave = an 8 x 8 pixel block
    r = rays[idx][valid[idx] == 1]
    sh = np.zeros((len(r), 8), np.float32)
    sh[:, :6] = r
    sh[:, 6] = 1e6
    f = {k: i for i, k in enumerate(A.FIELDS)}
    print(f"  the rays of bounce {bounce} as any-hit walks (max_t = 1e6), per visiting order:")
    for oname, code in (("near child first (the reference's)", 0), ("larger box first", 20), ("more opaque child first (rule 1)", 21), ("smaller subtree first (rule 2)", 22),
                        ("more opaque per node first (rule 3)", 23), ("opaque area per node first", 24)):
        acc = np.zeros(24, np.float64)
        for at in range(0, len(sh) - 511, 512):
            span = np.ascontiguousarray(sh[at:at + 512])
            out = np.zeros(24, np.uint64)
            hit = np.zeros(512, np.uint8)
            sim.sim_wave(C.byref(sc), _p(span), C.c_uint32(512), code, 16, 16, _p(out), _p(hit))
            acc += out
        n = acc[f["rays"]]
        print(f"      {oname:38s} node visits per ray that hits {acc[f['visits_occluded']] / max(acc[f['occluded']], 1):6.2f}   inner trips per ray {acc[f['inner_trips']] / n:.3f}   "
              f"leaf trips {acc[f['leaf_trips']] / n:.3f}   triangle iterations {acc[f['leaf_iters']] / n:.3f}   lanes per inner trip {acc[f['inner_lanes']] / max(acc[f['inner_trips']], 1):.1f}")


if __name__ == "__main__":
    main()
