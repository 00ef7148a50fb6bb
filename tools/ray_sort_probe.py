#!/usr/bin/env python3
"""What would sorting the bounce rays of a large scene buy?  The PRODUCTION nearest-hit walk (rpt_debug_trace_rays_production: the stage rpt_render
launches for the scene) on the same N diffuse-bounce-like rays — origins on random triangles, directions cosine-distributed about the face normal —
in three slot orders: random (what a bounce leaves in the slots: neighbours in the image, strangers in the scene), sorted by the Morton code of
the origin (30 bits), sorted by direction octant then origin, and bucketed by a 12-bit origin cell only.  Three launches per order; the walk's
durations come from the kernel trace: tools/ray_sort_probe.sh [deepbvh|scatter] [million rays] runs this under rocprofv3 and prints them."""
import importlib
import os
import sys

import numpy as np

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
rpt = importlib.import_module("rust-path-tracer_amd"); hip = importlib.import_module("rust-path-tracer_amd.hip")
from scenes import deep_bvh_scene, scatter_scene  # noqa: E402


def morton30(p, lo, hi):
    q = np.clip(((p - lo) / (hi - lo) * 1024.0).astype(np.int64), 0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "deepbvh"
    n = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 8_000_000
    w = deep_bvh_scene(1_000_000) if kind == "deepbvh" else scatter_scene(1_000_000)
    rng = np.random.default_rng(17)
    v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)[:, :3]
    t = w.indices
    pick = rng.integers(0, len(t), n)
    a, b, c = v[t["v0"][pick]], v[t["v1"][pick]], v[t["v2"][pick]]
    r1, r2 = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    s = np.sqrt(r1)
    p = (1 - s)[:, None] * a + (s * (1 - r2))[:, None] * b + (s * r2)[:, None] * c
    nrm = np.cross(b - a, c - a)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    nrm *= np.where(rng.random(n) < 0.5, -1.0, 1.0)[:, None].astype(np.float32)
    # cosine hemisphere about nrm
    u1, u2 = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    rr, phi = np.sqrt(u1), 2 * np.pi * u2
    helper = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    tx = np.cross(nrm, helper); tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(nrm, tx)
    d = (rr * np.cos(phi))[:, None] * tx + (rr * np.sin(phi))[:, None] * ty + np.sqrt(np.maximum(0, 1 - u1))[:, None] * nrm
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = (p + 1e-3 * nrm).astype(np.float32)
    lo, hi = v.min(0), v.max(0)
    m = morton30(o, lo, hi)
    octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
    orders = {"random (as a bounce leaves them)": np.arange(n), "origin Morton code": np.argsort(m, kind="stable"),
              "direction octant, then origin": np.argsort((octant << 30) | m, kind="stable"),
              "origin cell of 1/16 of the extent (12 bits), else random": np.argsort(morton30(o, lo, hi) >> 18, kind="stable")}
    side = 1
    while side * side * 8 < n:
        side *= 2
    cfg = rpt.default_config(side, side)
    r = hip.Renderer(0)
    r.set_samples_in_flight(8)
    r.upload_scene(w); r.set_config(cfg); r.reset(rpt.blue_noise_seeds(side, side))
    ref = None
    for name, idx in orders.items():
        for rep in range(3):
            tt, tri, fl = r.debug_trace_rays_production(o[idx], d[idx])
        back = np.empty(n, np.int64); back[idx] = np.arange(n)
        res = (tt[back].tobytes(), tri[back].tobytes(), fl[back].tobytes())
        if ref is None:
            ref = res
        assert res == ref, "the order of the slots changed a result"
        print(f"ORDER {name} | {kind}, {n / 1e6:.1f} M rays, hits {int((fl & 1).sum())}", flush=True)
    r.close()


if __name__ == "__main__":
    main()
