#!/usr/bin/env python3
"""Analysis tool (CPU only): how well does a wave64 keep its lanes busy while traversing?

Uses the oracle's event log (per ray: the sequence of inner-node / leaf visits of the reference traversal) on
rays dumped from real paths, groups rays into waves the way the GPU does (8x8 pixel blocks) and replays loop
structures as SIMT schedules.  Output: steps per wave and lane utilisation for each structure.
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
hip = importlib.import_module("rust-path-tracer_amd.hip")

COST_INNER, COST_LEAF = 100.0, 75.0      # VALU instructions per step (from the ISA)


def events_for(orc, sc, o, d, cap=256):
    n = len(o)
    ev = np.zeros((n, cap), np.uint8)
    ln = np.zeros(n, np.uint32)
    orc.lib.oracle_trace_events(C.byref(sc), C.c_size_t(n), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                ev.ctypes.data_as(C.c_void_p), C.c_uint32(cap), ln.ctypes.data_as(C.c_void_p))
    assert ln.max() <= cap
    return ev, ln


def sim_while_while(ev, ln):
    """All lanes walk inner nodes until each stands on a leaf (or is done), then leaves are tested together."""
    pos = np.zeros(len(ln), np.int64)
    inner_steps = leaf_steps = 0
    inner_lane = leaf_lane = 0
    while np.any(pos < ln):
        while True:
            at_inner = (pos < ln) & (ev[np.arange(len(ln)), np.minimum(pos, ev.shape[1] - 1)] == 0)
            if not at_inner.any():
                break
            inner_steps += 1
            inner_lane += int(at_inner.sum())
            pos[at_inner] += 1
        at_leaf = pos < ln
        if at_leaf.any():
            leaf_steps += 1
            leaf_lane += int(at_leaf.sum())
            pos[at_leaf] += 1
    return inner_steps, leaf_steps, inner_lane, leaf_lane


def sim_if_if(ev, ln):
    """One loop, each iteration every live lane takes ONE step; both bodies are issued when both kinds are present."""
    pos = np.zeros(len(ln), np.int64)
    inner_steps = leaf_steps = inner_lane = leaf_lane = 0
    while np.any(pos < ln):
        live = pos < ln
        kind = ev[np.arange(len(ln)), np.minimum(pos, ev.shape[1] - 1)]
        a = live & (kind == 0)
        b = live & (kind == 1)
        if a.any():
            inner_steps += 1
            inner_lane += int(a.sum())
        if b.any():
            leaf_steps += 1
            leaf_lane += int(b.sum())
        pos[live] += 1
    return inner_steps, leaf_steps, inner_lane, leaf_lane


def report(name, results, n_rays):
    i_s = sum(r[0] for r in results); l_s = sum(r[1] for r in results)
    i_l = sum(r[2] for r in results); l_l = sum(r[3] for r in results)
    cost = i_s * COST_INNER + l_s * COST_LEAF
    useful = (i_l * COST_INNER + l_l * COST_LEAF) / 64.0
    print(f"  {name:28s} inner steps/wave {i_s/len(results):6.1f} leaf steps/wave {l_s/len(results):5.1f}  "
          f"lane util inner {i_l/(64*max(i_s,1)):.2f} leaf {l_l/(64*max(l_s,1)):.2f}  overall {useful/cost:.2f}  "
          f"cost/ray {cost/n_rays:7.1f}")


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)          # slot order: 8x8 blocks
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    for bounce in range(4):
        rays = np.zeros((W * H, 6), np.float32)
        valid = np.zeros(W * H, np.uint8)
        orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                                 rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
        rays, valid = rays[px], valid[px]
        o = np.ascontiguousarray(rays[:, :3]); d = np.ascontiguousarray(rays[:, 3:])
        ev, ln = events_for(orc, sc, o, d)
        ln = np.where(valid == 1, ln, 0)
        n_rays = int(valid.sum())
        print(f"{scene} bounce {bounce}: {n_rays} rays, node visits/ray mean {ln[valid==1].mean():.1f} max {ln.max()} "
              f"leaf visits/ray {ev[valid==1].sum(axis=1).mean():.2f}")
        waves = [(ev[i:i + 64], ln[i:i + 64]) for i in range(0, len(ln), 64)]
        waves = [wv for wv in waves if wv[1].max() > 0]
        report("while-while", [sim_while_while(*wv) for wv in waves], n_rays)
        report("if-if (one step per iter)", [sim_if_if(*wv) for wv in waves], n_rays)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("refill", "threshold", "sort", "regroup", "stream", "plateau")):
    main()


def sim_refill(ev, ln, valid, chunk, thresh, cost_refill=60.0, mode="ifif"):
    """Persistent lanes: a wave owns `chunk` consecutive rays; idle lanes are refilled when >= thresh are idle.
    Returns total cost (instruction issue slots) for all rays."""
    total = 0.0
    n = len(ln)
    for base in range(0, n, chunk):
        idx = [i for i in range(base, min(base + chunk, n)) if valid[i]]
        nxt = 0
        lane_ray = [-1] * 64
        lane_pos = [0] * 64
        while True:
            idle = [l for l in range(64) if lane_ray[l] < 0]
            if nxt < len(idx) and (len(idle) >= thresh or len(idle) == 64):
                for l in idle:
                    if nxt < len(idx):
                        lane_ray[l] = idx[nxt]; lane_pos[l] = 0; nxt += 1
                total += cost_refill
            live = [l for l in range(64) if lane_ray[l] >= 0]
            if not live:
                break
            if mode == "ifif":
                kinds = [ev[lane_ray[l], lane_pos[l]] for l in live]
                if 0 in kinds: total += COST_INNER
                if 1 in kinds: total += COST_LEAF
                for l in live:
                    lane_pos[l] += 1
                    if lane_pos[l] >= ln[lane_ray[l]]: lane_ray[l] = -1
            else:   # while-while
                while True:
                    inner = [l for l in live if lane_ray[l] >= 0 and ev[lane_ray[l], lane_pos[l]] == 0]
                    if not inner: break
                    total += COST_INNER
                    for l in inner:
                        lane_pos[l] += 1
                        if lane_pos[l] >= ln[lane_ray[l]]: lane_ray[l] = -1
                leaf = [l for l in live if lane_ray[l] >= 0]
                if leaf:
                    total += COST_LEAF
                    for l in leaf:
                        lane_pos[l] += 1
                        if lane_pos[l] >= ln[lane_ray[l]]: lane_ray[l] = -1
    return total


def main_refill():
    scene = sys.argv[2] if len(sys.argv) > 2 else "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    bounce = 2
    rays = np.zeros((W * H, 6), np.float32)
    valid = np.zeros(W * H, np.uint8)
    orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                             rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
    rays, valid = rays[px], valid[px]
    ev, ln = events_for(orc, sc, np.ascontiguousarray(rays[:, :3]), np.ascontiguousarray(rays[:, 3:]))
    ln = np.where(valid == 1, ln, 0)
    sub = slice(0, 4096)
    evs, lns, vs = ev[sub], ln[sub], valid[sub]
    n_rays = int(vs.sum())
    ideal = (float((evs[:, :] == 0)[np.arange(len(lns))[:, None], :][..., :0].sum()))  # placeholder
    inner_total = sum(int((evs[i, :lns[i]] == 0).sum()) for i in range(len(lns)))
    leaf_total = sum(int((evs[i, :lns[i]] == 1).sum()) for i in range(len(lns)))
    print(f"{scene} bounce {bounce}: ideal cost/ray {(inner_total*COST_INNER+leaf_total*COST_LEAF)/64/n_rays:.1f}")
    for mode in ("ifif", "whilewhile"):
        for chunk, thresh in [(64, 64), (128, 16), (256, 16), (256, 8), (256, 32), (512, 16), (1024, 16)]:
            c = sim_refill(evs, lns, vs, chunk, thresh, mode=mode)
            print(f"  {mode:10s} chunk {chunk:5d} refill>= {thresh:2d}: cost/ray {c/n_rays:7.1f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "refill":
    main_refill()


def sim_threshold(ev, ln, valid, chunk, refill_thresh, leaf_k, cost_refill=60.0):
    """if-if with deferred leaves: lanes standing on a leaf wait until >= leaf_k lanes wait (or nobody is at an inner
    node); lanes at inner nodes step every iteration.  Optional refill from a chunk (chunk == 64: none)."""
    total = 0.0
    n = len(ln)
    for base in range(0, n, chunk):
        idx = [i for i in range(base, min(base + chunk, n)) if valid[i]]
        nxt = 0
        lane_ray = [-1] * 64
        lane_pos = [0] * 64
        while True:
            idle = [l for l in range(64) if lane_ray[l] < 0]
            if nxt < len(idx) and (len(idle) >= refill_thresh or len(idle) == 64):
                for l in idle:
                    if nxt < len(idx):
                        lane_ray[l] = idx[nxt]; lane_pos[l] = 0; nxt += 1
                total += cost_refill
            live = [l for l in range(64) if lane_ray[l] >= 0]
            if not live:
                break
            inner = [l for l in live if ev[lane_ray[l], lane_pos[l]] == 0]
            leaf = [l for l in live if ev[lane_ray[l], lane_pos[l]] == 1]
            step = []
            if inner:
                total += COST_INNER
                step += inner
            if leaf and (len(leaf) >= leaf_k or not inner):
                total += COST_LEAF
                step += leaf
            for l in step:
                lane_pos[l] += 1
                if lane_pos[l] >= ln[lane_ray[l]]: lane_ray[l] = -1
    return total


def main_threshold():
    scene = sys.argv[2] if len(sys.argv) > 2 else "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    for bounce in (0, 2):
        rays = np.zeros((W * H, 6), np.float32)
        valid = np.zeros(W * H, np.uint8)
        orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                                 rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
        rays, valid = rays[px], valid[px]
        ev, ln = events_for(orc, sc, np.ascontiguousarray(rays[:, :3]), np.ascontiguousarray(rays[:, 3:]))
        ln = np.where(valid == 1, ln, 0)
        sub = slice(0, 4096)
        evs, lns, vs = ev[sub], ln[sub], valid[sub]
        n_rays = int(vs.sum())
        print(f"{scene} bounce {bounce}")
        for chunk, rt in [(64, 64), (256, 16)]:
            for k in (1, 4, 8, 12, 16, 24, 32, 64):
                c = sim_threshold(evs, lns, vs, chunk, rt, k)
                print(f"  chunk {chunk:4d} leaf_k {k:2d}: cost/ray {c/n_rays:7.1f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "threshold":
    main_threshold()


def main_sort():
    """Does regrouping the rays of a 512-thread workgroup by direction (octant / finer) help a wave's coherence?"""
    scene = sys.argv[2] if len(sys.argv) > 2 else "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    bounce = 2
    rays = np.zeros((W * H, 6), np.float32)
    valid = np.zeros(W * H, np.uint8)
    orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                             rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
    rays, valid = rays[px], valid[px]
    ev, ln = events_for(orc, sc, np.ascontiguousarray(rays[:, :3]), np.ascontiguousarray(rays[:, 3:]))
    ln = np.where(valid == 1, ln, 0)
    n = 8192
    d = rays[:n, 3:]
    for name, keyfn in [("none", None),
                        ("octant", lambda d: (d[:, 0] > 0) * 4 + (d[:, 1] > 0) * 2 + (d[:, 2] > 0)),
                        ("octant+major axis", lambda d: ((d[:, 0] > 0) * 4 + (d[:, 1] > 0) * 2 + (d[:, 2] > 0)) * 3 + np.argmax(np.abs(d), axis=1)),
                        ("path length (oracle knowledge)", "len")]:
        for group in (512, 2048):
            perm = np.arange(n)
            if keyfn is not None:
                for b in range(0, n, group):
                    sl = slice(b, b + group)
                    key = ln[sl] if keyfn == "len" else keyfn(d[sl])
                    perm[sl] = b + np.argsort(key, kind="stable")
            c = sim_threshold(ev[perm], ln[perm], valid[:n][perm], 64, 64, 8)
            print(f"  regroup by {name:32s} within {group:5d}: cost/ray {c/int(valid[:n].sum()):7.1f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sort":
    main_sort()


def sim_regroup(ev, ln, valid, group, budgets, leaf_k, cost_inner, cost_leaf, cost_regroup):
    """Workgroup-level regrouping: the `group` rays of a workgroup start one per lane; after budgets[0] loop trips every
    wave stops, the unfinished rays are re-dealt densely to lanes (state handed over), and the new waves run
    budgets[1] more trips, ...; the last budget repeats.  budgets = [10**9] is the plain deferred-leaf loop."""
    total = 0.0
    n = len(ln)
    for base in range(0, n, group):
        rays = [(i, 0) for i in range(base, min(base + group, n)) if valid[i] and ln[i] > 0]
        phase = 0
        while rays:
            budget = budgets[min(phase, len(budgets) - 1)]
            nxt = []
            for w0 in range(0, len(rays), 64):
                lanes = [list(r) for r in rays[w0:w0 + 64]]
                trips = 0
                while trips < budget:
                    live = [l for l in lanes if l[1] < ln[l[0]]]
                    if not live:
                        break
                    inner = [l for l in live if ev[l[0], l[1]] == 0]
                    leaf = [l for l in live if ev[l[0], l[1]] == 1]
                    if inner:
                        total += cost_inner
                        for l in inner: l[1] += 1
                    if leaf and (len(leaf) >= leaf_k or not inner):
                        total += cost_leaf
                        for l in leaf: l[1] += 1
                    trips += 1
                left = [tuple(l) for l in lanes if l[1] < ln[l[0]]]
                if left:
                    total += cost_regroup
                nxt += left
            rays = nxt
            phase += 1
    return total


def main_regroup():
    scene = sys.argv[2] if len(sys.argv) > 2 else "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    CI, CL, CR = 214.0, 200.0, 150.0       # SIMD cycles per inner step / leaf step / hand-over (per wave)
    for bounce in (0, 2):
        rays = np.zeros((W * H, 6), np.float32)
        valid = np.zeros(W * H, np.uint8)
        orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                                 rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
        rays, valid = rays[px], valid[px]
        ev, ln = events_for(orc, sc, np.ascontiguousarray(rays[:, :3]), np.ascontiguousarray(rays[:, 3:]))
        ln = np.where(valid == 1, ln, 0)
        sub = slice(0, 8192)
        evs, lns, vs = ev[sub], ln[sub], valid[sub]
        n_rays = int(vs.sum())
        print(f"{scene} bounce {bounce}: visits/ray mean {lns[vs==1].mean():.1f} p50 {np.percentile(lns[vs==1],50):.0f} p90 {np.percentile(lns[vs==1],90):.0f} max {lns.max()}")
        for name, budgets in [("no regroup", [10**9]), ("16,8..", [16, 8]), ("24,8..", [24, 8]), ("24,12..", [24, 12]), ("32,8..", [32, 8]),
                              ("32,16..", [32, 16]), ("20,10,10,20..", [20, 10, 10, 20]), ("40,..", [40, 16]), ("12,12..", [12]), ("8,8..", [8])]:
            c = sim_regroup(evs, lns, vs, 1024, budgets, 16, CI, CL, CR)
            print(f"  budgets {name:16s}: cycles/ray {c/n_rays:7.1f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "regroup":
    main_regroup()


def sim_stream(ev, ln, valid, rays_per_lane, trips, refill_thresh, leaf_k, cost_inner, cost_leaf, cost_refill):
    """k_traverse_nearest_stream as built: a wave owns rays_per_lane*64 consecutive slots; every `trips` loop trips, if at
    least refill_thresh lanes are idle and slots remain, the idle lanes take the next slots (non-pending slots leave a lane
    idle until the next look)."""
    total = 0.0
    n = len(ln)
    span = rays_per_lane * 64
    for base in range(0, n, span):
        idx = list(range(base, min(base + span, n)))
        nxt = 0
        lane_ray = [-1] * 64
        lane_pos = [0] * 64
        while True:
            idle = [l for l in range(64) if lane_ray[l] < 0]
            if nxt < len(idx) and len(idle) >= refill_thresh:
                for l in idle:
                    if nxt < len(idx):
                        i = idx[nxt]; nxt += 1
                        if valid[i] and ln[i] > 0:
                            lane_ray[l] = i; lane_pos[l] = 0
                total += cost_refill
                continue
            if len(idle) == 64:
                break
            budget = trips if nxt < len(idx) else 10**9
            t = 0
            while t < budget:
                live = [l for l in range(64) if lane_ray[l] >= 0]
                if not live:
                    break
                inner = [l for l in live if ev[lane_ray[l], lane_pos[l]] == 0]
                leaf = [l for l in live if ev[lane_ray[l], lane_pos[l]] == 1]
                step = []
                if inner:
                    total += cost_inner; step += inner
                if leaf and (len(leaf) >= leaf_k or not inner):
                    total += cost_leaf; step += leaf
                for l in step:
                    lane_pos[l] += 1
                    if lane_pos[l] >= ln[lane_ray[l]]: lane_ray[l] = -1
                t += 1
    return total


def main_stream():
    scene = sys.argv[2] if len(sys.argv) > 2 else "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    CI, CL, CR = 214.0, 200.0, 300.0
    for bounce in (0, 2):
        rays = np.zeros((W * H, 6), np.float32)
        valid = np.zeros(W * H, np.uint8)
        orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                                 rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
        rays, valid = rays[px], valid[px]
        ev, ln = events_for(orc, sc, np.ascontiguousarray(rays[:, :3]), np.ascontiguousarray(rays[:, 3:]))
        ln = np.where(valid == 1, ln, 0)
        sub = slice(0, 8192)
        evs, lns, vs = ev[sub], ln[sub], valid[sub]
        n_rays = int(vs.sum())
        print(f"{scene} bounce {bounce}")
        for name, (r, t, f) in [("one ray per lane", (1, 10**9, 64)), ("R4 T12 F16", (4, 12, 16)), ("R8 T12 F16", (8, 12, 16)), ("R4 T4 F8", (4, 4, 8)),
                                ("R4 T1 F1", (4, 1, 1)), ("R16 T4 F8", (16, 4, 8)), ("R128 T1 F1", (128, 1, 1))]:
            c = sim_stream(evs, lns, vs, r, t, f, 16, CI, CL, CR)
            print(f"  {name:18s}: cycles/ray {c/n_rays:7.1f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stream":
    main_stream()


def main_plateau():
    """Round-2 analysis behind DESIGN.md §8 item 2: the streamed walk with one body per trip (the kernel as built: 8 trips
    between looks, refill at >= 12 idle lanes, costs 75 / 70 / 120 VALU per inner / leaf / refill step) replayed on real
    bounce-1 / bounce-2 rays of DarkCornell — (a) dealt in slot order, (b) dealt sorted by direction octant / direction
    cell / origin cell / TRUE walk length inside 4 096-slot spans, (c) with 1, 2 or 3 rays per lane."""
    scene = "DarkCornell"
    W = H = 128
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    order = hip.tile_order(W, H, 0, 1)
    px = (order >> 16).astype(np.int64) * W + (order & 0xFFFF).astype(np.int64)
    CI,CL=75.0,70.0
    def sim_g(ev,ln,valid,R,T,F,CR,alpha=1.0):
        tot=dict(inner=0.0,leaf=0.0,refill=0.0,inner_l=0,leaf_l=0,inner_s=0,leaf_s=0)
        n=len(ln); span=R*64
        for base in range(0,n,span):
            idx=[i for i in range(base,min(base+span,n))]
            nxt=0; ray=[-1]*64; pos=[0]*64
            while True:
                idle=[l for l in range(64) if ray[l]<0]
                if nxt<len(idx) and len(idle)>=F:
                    for l in idle:
                        if nxt<len(idx):
                            i=idx[nxt]; nxt+=1
                            if valid[i] and ln[i]>0: ray[l]=i; pos[l]=0
                    tot['refill']+=CR; continue
                if len(idle)==64: break
                budget=T if nxt<len(idx) else 10**9
                t=0
                while t<budget:
                    live=[l for l in range(64) if ray[l]>=0]
                    if not live: break
                    inner=[l for l in live if ev[ray[l],pos[l]]==0]
                    leaf=[l for l in live if ev[ray[l],pos[l]]==1]
                    if len(leaf)>alpha*len(inner): step=leaf; tot['leaf']+=CL; tot['leaf_l']+=len(leaf); tot['leaf_s']+=1
                    else: step=inner; tot['inner']+=CI; tot['inner_l']+=len(inner); tot['inner_s']+=1
                    for l in step:
                        pos[l]+=1
                        if pos[l]>=ln[ray[l]]: ray[l]=-1
                    t+=1
        return tot
    for bounce in (1,2):
        rays=np.zeros((W*H,6),np.float32); valid=np.zeros(W*H,np.uint8)
        orc.lib.oracle_dump_rays(C.byref(cfg),C.byref(sc),seeds.ctypes.data_as(C.c_void_p),C.c_uint32(bounce),rays.ctypes.data_as(C.c_void_p),valid.ctypes.data_as(C.c_void_p))
        rays,valid=rays[px],valid[px]
        ev,ln=events_for(orc,sc,np.ascontiguousarray(rays[:,:3]),np.ascontiguousarray(rays[:,3:]))
        ln=np.where(valid==1,ln,0)
        N=16384
        evs,lns,vs,rs=ev[:N],ln[:N],valid[:N],rays[:N]; n=int(vs.sum())
        ninner=sum(int((evs[i,:lns[i]]==0).sum()) for i in range(N)); nleaf=sum(int((evs[i,:lns[i]]==1).sum()) for i in range(N))
        ideal=(ninner*CI+nleaf*CL)/64/n
        def run(name,perm):
            t=sim_g(evs[perm],lns[perm],vs[perm],64,8,12,120,1.0)
            c=(t['inner']+t['leaf']+t['refill'])/n
            print(f"  bounce {bounce} {name:34s} cyc/ray {c:6.1f} util {ideal/c*100:5.1f}% inner lanes {t['inner_l']/max(t['inner_s'],1):5.1f} leaf lanes {t['leaf_l']/max(t['leaf_s'],1):5.1f}")
        ident=np.arange(N)
        run("slot order",ident)
        d=rs[:,3:6]; o=rs[:,:3]
        octant=(d[:,0]>0)*1+(d[:,1]>0)*2+(d[:,2]>0)*4
        span=4096
        def sort_in_spans(key):
            perm=np.concatenate([b+np.argsort(key[b:b+span],kind='stable') for b in range(0,N,span)])
            return perm
        run("octant within 4096",sort_in_spans(octant))
        # finer: octahedral 8x8 direction cell
        ad=np.abs(d).sum(1,keepdims=True); pxy=d[:,:2]/ad
        neg=d[:,2]<0
        ox=np.where(neg,(1-np.abs(pxy[:,1]))*np.sign(pxy[:,0]),pxy[:,0]); oy=np.where(neg,(1-np.abs(pxy[:,0]))*np.sign(pxy[:,1]),pxy[:,1])
        cx=np.clip(((ox+1)*4).astype(int),0,7); cy=np.clip(((oy+1)*4).astype(int),0,7)
        run("dir cell 8x8 within 4096",sort_in_spans(cx*8+cy))
        lo=o.min(0); hi=o.max(0); oc=np.clip(((o-lo)/(hi-lo+1e-6)*4).astype(int),0,3)
        okey=oc[:,0]*16+oc[:,1]*4+oc[:,2]
        run("origin cell 4^3 then dir 8x8",sort_in_spans(okey*64+cx*8+cy))
        run("dir 8x8 then origin 4^3",sort_in_spans((cx*8+cy)*64+okey))
        run("walk length (oracle knowledge)",sort_in_spans(lns))
    def sim_sets(ev,ln,valid,CR,K=2):
        """K ray SETS per wave, no per-lane switch: register set k of lane l holds one ray; a trip runs ONE body on ONE set
        (K x 2 candidates), the one with the most lanes ready — what duplicated loop bodies over two register contexts
        would do, without the select cost of sim_2."""
        idx=np.nonzero(valid&(ln>0))[0]
        total=0; pool=0
        cur=np.full((64,K),-1); pos=np.zeros((64,K),int)
        def refill():
            nonlocal pool
            free=np.argwhere(cur<0)
            take=min(len(free),len(idx)-pool)
            for (l,k),r in zip(free[:take],idx[pool:pool+take]):
                cur[l,k]=r; pos[l,k]=0
            pool+=take
        refill()
        isum=lsum=0; ic=lc=0
        trips=0
        while (cur>=0).any():
            act=cur>=0
            kinds=np.where(act, ev[np.maximum(cur,0),np.minimum(pos,ev.shape[1]-1)],255)
            cnt=[((kinds[:,k]==0).sum(),(kinds[:,k]==1).sum()) for k in range(K)]
            k=max(range(K),key=lambda q:max(cnt[q]))
            leaf=cnt[k][1]>cnt[k][0]
            adv=kinds[:,k]==(1 if leaf else 0)
            if leaf: lsum+=adv.sum(); lc+=1
            else: isum+=adv.sum(); ic+=1
            total+=CL if leaf else CI
            pos[:,k]=np.where(adv,pos[:,k]+1,pos[:,k])
            done=act&(pos>=ln[np.maximum(cur,0)])
            cur[done]=-1
            trips+=1
            if trips%8==0 and pool<len(idx) and (cur<0).sum()>=12*K:
                refill(); total+=CR
        return total, isum/max(ic,1), lsum/max(lc,1)
    def sim_spec(ev,ln,valid,CR,CA=15.0,CX=5.0,alpha=1.0):
        """One ray per lane, but a lane waiting on a LEAF whose next node (the stack top: known before the leaf is tested) is an
        inner node runs the box tests of THAT node during an inner trip and keeps (t_left, t_right, two predicate bits); when its
        leaf trip comes it tests the leaf, then applies the kept result against the updated best t (CA instructions) and stands
        on the chosen child — two steps of the ray in one inner + one leaf trip.  Exact: only the comparison with the best t
        depends on the leaf's outcome, and it is made afterwards."""
        idx=np.nonzero(valid&(ln>0))[0]
        total=0.0; pool=0
        cur=np.full(64,-1); pos=np.zeros(64,int); spec=np.zeros(64,bool)
        L=ev.shape[1]
        def refill():
            nonlocal pool
            free=np.nonzero(cur<0)[0]
            take=min(len(free),len(idx)-pool)
            cur[free[:take]]=idx[pool:pool+take]; pos[free[:take]]=0; spec[free[:take]]=False
            pool+=take
        refill()
        isum=lsum=ic=lc=0; trips=0; useful=0
        while (cur>=0).any():
            act=cur>=0
            c=np.maximum(cur,0)
            kind=np.where(act,ev[c,np.minimum(pos,L-1)],255)
            nxt_inner=act&(kind==1)&(pos+1<ln[c])&(ev[c,np.minimum(pos+1,L-1)]==0)&~spec
            n_in=(kind==0).sum(); n_leaf=(kind==1).sum(); n_sp=nxt_inner.sum()
            if n_leaf>alpha*(n_in+n_sp) or (n_in+n_sp)==0:
                adv=kind==1
                total+=CL+(CA if (adv&spec).any() else 0.0)
                two=adv&spec
                pos[adv]+=1; pos[two]+=1; spec[two]=False
                lsum+=adv.sum(); lc+=1
            else:
                adv=kind==0
                total+=CI+(CX if n_sp else 0.0)
                pos[adv]+=1; spec[nxt_inner]=True
                isum+=adv.sum()+n_sp; ic+=1
            done=act&(pos>=ln[c])
            cur[done]=-1
            trips+=1
            if trips%8==0 and pool<len(idx) and (cur<0).sum()>=12:
                refill(); total+=CR
        return total, isum/max(ic,1), lsum/max(lc,1)
    print("---- speculative box tests of the stack top by lanes that wait on a leaf")
    for alpha in (1.0,0.8,0.6):
        t,il,ll=sim_spec(evs,lns,vs==1,120.0,alpha=alpha)
        print(f"  bounce {bounce} alpha {alpha}: cyc/ray {t/n:6.1f} util(vs plain ideal) {ideal/(t/n)*100:5.1f}% lanes busy in inner trips {il:5.1f} in leaf trips {ll:5.1f}")
    print("---- K register sets per wave, one body on one set per trip (no per-lane switch)")
    for K in (1,2,3,4):
        t,il,ll=sim_sets(evs,lns,vs==1,120.0,K)
        print(f"  bounce {bounce} K={K} sets: cyc/ray {t/n:6.1f} util {ideal/(t/n)*100:5.1f}% inner lanes {il:5.1f} leaf lanes {ll:5.1f}")
    print("---- two rays per lane")
    def sim_2(ev,ln,valid,CR,K=2):
        n=len(ln); tot=0.0; inner_l=leaf_l=inner_s=leaf_s=0
        nxt=0; ray=[[-1]*K for _ in range(64)]; pos=[[0]*K for _ in range(64)]
        def refill():
            nonlocal nxt,tot
            cnt=0
            for l in range(64):
                for k in range(K):
                    if ray[l][k]<0 and nxt<n:
                        while nxt<n and not (valid[nxt] and ln[nxt]>0): nxt+=1
                        if nxt<n: ray[l][k]=nxt; pos[l][k]=0; nxt+=1; cnt+=1
            if cnt: tot+=CR
        trips=0
        while True:
            nidle=sum(1 for l in range(64) for k in range(K) if ray[l][k]<0)
            if nxt<n and nidle>=12*K and trips%8==0: refill()
            live=[(l,k) for l in range(64) for k in range(K) if ray[l][k]>=0]
            if not live:
                if nxt>=n: break
                refill(); continue
            inner_lanes={}; leaf_lanes={}
            for l,k in live:
                t=ev[ray[l][k],pos[l][k]]
                (inner_lanes if t==0 else leaf_lanes).setdefault(l,k)
            if len(leaf_lanes)>len(inner_lanes): step=leaf_lanes; tot+=CL; leaf_l+=len(step); leaf_s+=1
            else: step=inner_lanes; tot+=CI; inner_l+=len(step); inner_s+=1
            for l,k in step.items():
                pos[l][k]+=1
                if pos[l][k]>=ln[ray[l][k]]: ray[l][k]=-1
            trips+=1
        return tot,inner_l/max(inner_s,1),leaf_l/max(leaf_s,1)
    for K in (1,2,3):
        t,il,ll=sim_2(evs,lns,vs,120,K)
        print(f"  bounce {bounce} K={K} rays/lane: cyc/ray {t/n:6.1f} util {ideal/(t/n)*100:5.1f}% inner lanes {il:5.1f} leaf lanes {ll:5.1f}")



if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "plateau":
    main_plateau()
