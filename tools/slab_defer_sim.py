#!/usr/bin/env python3
"""Deferred-exact slab test for the LDS walk: would THREE bodies per trip pay?  (CPU analysis, no GPU.)

Round 3 rejected a floating-point filter for the box test because a wave would run the exact path whenever ANY of its lanes had an
undecided comparison (tools/slab_filter_sim.py: 3.9 % of the visits -> 75 % of the trips).  The walk already knows a better rule for
leaves: a lane that needs the other body PARKS on its node until that body is issued for enough lanes.  This tool replays that rule:

  filter body   t~ = fma(p, 1/d, -(o * 1/d)) per plane (1 VALU instead of sub + Markstein's 3), max3 / min3, the reference's
                comparisons on t~, and ONE rigorous per-visit bound E: a lane whose every comparison is further from equality than
                the bound takes the decision (hit_l, hit_r, swap) — the very decision the exact arithmetic takes — and steps on;
                a lane with a comparison inside the bound parks as "needs exact" on the same node;
  exact body    today's inner step (slab_pair_lds<true>), issued for the parked lanes only;
  leaf body     unchanged.

Per trip ONE body runs, the one with the most lanes ready (the kernel's RPT_LEAF_GREEDY_PCT = 100 rule, extended to three).  Costs in
VALU wave-instructions per step (ISA counts of the built kernel): exact inner 75, leaf 70, refill look 120; filter inner 75 - 36
(eleven... twelve planes x 3 instructions less) + the bound test, a parameter (48 ... 55).

Replayed over real DarkCornell rays of all four bounces (dumped by the oracle, slot order = 8 x 8 pixel blocks per wave), streamed as
k_traverse_nearest_stream deals them (spans of R rays per lane, a refill look every T trips when >= F lanes are idle).

The bound.  With ird = RN(1/d), noi = RN(-o * ird):  t~ = RN(p * ird + noi) differs from the reference's t = RN(RN(p - o) / d) by at most
u (4 |p ird| + 5 |o ird|) (1 + O(u)),  u = 2^-24.  Three ways to turn that into ONE number per visit, cheapest first:
  ray    E = 6 u (Pmax max|ird| + max|oi|)            Pmax = largest |coordinate| of any box of the scene: a per-ray constant
  pair   E = 6 u (m_pair max|ird| + max|oi|)          m_pair = largest |coordinate| of the two boxes (one more LDS word + 1 fma per visit)
  plane  the per-plane bound of slab_filter_sim.py    (not implementable at 1 VALU per plane; the floor)
usage: python tools/slab_defer_sim.py [scene] [size]
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
U = 2.0 ** -24
MODES = ("ray", "pair", "plane", "pair+dom", "pair+dom+flat", "ideal")


def walk(nodes_f, nodes_u, tri_test, o, d, pmax):
    """Lock-step replay of intersect_front_to_back over all rays (float32, the reference's operations).  Returns per ray the event
    sequence kind[i, k] (0 inner visit, 1 leaf visit) with its length, and per inner visit and bound mode whether the filter could NOT
    decide it (amb[mode][i, k]), in the cheap form (min3 of the three distances against 2E) and the short-circuit form."""
    n = len(o)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ird = (np.float32(1.0) / d32).astype(np.float32)
        noi = (-(o32 * ird)).astype(np.float32)
    ird_m = np.abs(ird).max(axis=1).astype(np.float64)
    oi_m = np.abs(noi).max(axis=1).astype(np.float64)
    cap = 160
    kind = np.zeros((n, cap), np.uint8)
    amb = {m: np.zeros((n, cap), bool) for m in MODES}
    amb_sc = {m: np.zeros((n, cap), bool) for m in MODES}
    direct = np.zeros((n, cap), bool)
    ln = np.zeros(n, np.int64)
    cur = np.zeros(n, np.int64)
    sp = np.zeros(n, np.int64)
    stack = np.zeros((n, 40), np.int64)
    best = np.full(n, 1000000.0, np.float32)
    live = np.ones(n, bool)
    ties = both_hits = 0
    while live.any():
        cnt = nodes_u[cur, 3]
        inner = live & (cnt == 0)
        leaf = live & (cnt > 0)
        kind[leaf, ln[leaf]] = 1
        for i in np.flatnonzero(leaf):
            first = nodes_u[cur[i], 7]
            for k in range(cnt[i]):
                t = tri_test(first + k, o32[i], d32[i])
                if t is not None and t > np.float32(0.001) and t < best[i]:
                    best[i] = t
        pop = leaf.copy()
        if inner.any():
            idx = np.flatnonzero(inner)
            left = nodes_u[cur[idx], 7].astype(np.int64)
            oo, dd, ii, nn = o32[idx], d32[idx], ird[idx], noi[idx]
            bb = best[idx].astype(np.float64)
            m_pair = np.maximum(np.abs(nodes_f[left][:, [0, 1, 2, 4, 5, 6]]).max(axis=1), np.abs(nodes_f[left + 1][:, [0, 1, 2, 4, 5, 6]]).max(axis=1)).astype(np.float64)
            E = {"ray": 6 * U * (pmax * ird_m[idx] + oi_m[idx]), "pair": 6 * U * (m_pair * ird_m[idx] + oi_m[idx])}
            res = {}
            for side, child in (("l", left), ("r", left + 1)):
                lo = nodes_f[child, 0:3]
                hi = nodes_f[child, 4:7]
                with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                    t1 = ((lo - oo).astype(np.float32) / dd).astype(np.float32)
                    t2 = ((hi - oo).astype(np.float32) / dd).astype(np.float32)
                    a1 = (lo.astype(np.float64) * ii + nn).astype(np.float32)          # fma(p, ird, noi), rounded once
                    a2 = (hi.astype(np.float64) * ii + nn).astype(np.float32)
                    e1 = U * (4 * np.abs(lo.astype(np.float64) * ii) + 5 * np.abs(nn.astype(np.float64)))
                    e2 = U * (4 * np.abs(hi.astype(np.float64) * ii) + 5 * np.abs(nn.astype(np.float64)))
                neg = dd < 0                                                           # sign-selected near / far, as the plane records
                tn, tf = np.where(neg, t2, t1), np.where(neg, t1, t2)
                an, af = np.where(neg, a2, a1), np.where(neg, a1, a2)
                e_pl = np.maximum(e1, e2).max(axis=1)
                tmin, tmax = tn.max(axis=1), tf.min(axis=1)
                amin, amax = an.max(axis=1).astype(np.float64), af.min(axis=1).astype(np.float64)
                hit = (tmax >= tmin) & (tmax > 0) & (tmin < best[idx])
                res[side] = (hit, tmin, amin, amax, e_pl)
            hl, tl, aminl, amaxl, epl_l = res["l"]
            hr, tr, aminr, amaxr, epl_r = res["r"]
            E["plane"] = np.maximum(epl_l, epl_r)
            both = hl & hr
            both_hits += int(both.sum())
            ties += int((both & (tl == tr)).sum())
            # what-if variants on the per-pair bound:
            #  +dom   one bit per pair and direction octant: every near plane of L is <= the corresponding one of R in ray order, hence
            #         tl <= tr in the exact arithmetic (RN subtraction / division are monotone) and the children are certainly not swapped
            #  +flat  a pair with a flat child (a wall: near == far on an axis, where tmax == tmin bit for bit in any arithmetic) is marked at
            #         upload and goes to the exact body WITHOUT running the filter first (direct[] below)
            #  ideal  every exact equality (tie, flat axis, origin on a plane) counted as decided: the floor no flag scheme can beat
            E["pair+dom"] = E["pair+dom+flat"] = E["ideal"] = E["pair"]
            lo_l, hi_l, lo_r, hi_r = nodes_f[left, 0:3], nodes_f[left, 4:7], nodes_f[left + 1, 0:3], nodes_f[left + 1, 4:7]
            neg3 = dd < 0
            nl, nr_ = np.where(neg3, hi_l, lo_l), np.where(neg3, hi_r, lo_r)
            dom = np.where(neg3, nl >= nr_, nl <= nr_).all(axis=1)
            flat_pair = ((lo_l == hi_l) | (lo_r == hi_r)).any(axis=1)
            for m in MODES:
                e = E[m]
                a_box, a_box_sc, sure_hit = [], [], []
                for (amin, amax) in ((aminl, amaxl), (aminr, amaxr)):
                    d1, d2, d3 = amax - amin, amax, bb - amin                          # > 0 <=> the three conditions of intersect_aabb
                    cheap = np.minimum(np.minimum(np.abs(d1), np.abs(d2)), np.abs(d3)) <= 2 * e
                    s_miss = (d1 < -2 * e) | (d2 < -e) | (d3 < -e)
                    s_hit = (d1 > 2 * e) & (d2 > e) & (d3 > e)
                    a_box.append(cheap); a_box_sc.append(~(s_miss | s_hit)); sure_hit.append(s_hit)
                # the order of the children only matters when both are (certainly) hit
                c4 = np.abs(aminl - aminr) <= 2 * e
                if m in ("pair+dom", "pair+dom+flat"):
                    c4 = c4 & ~dom
                if m == "ideal":
                    c4 = c4 & (tl != tr)
                    for bi, (amin, amax) in enumerate(((aminl, amaxl), (aminr, amaxr))):
                        d1, d2, d3 = amax - amin, amax, bb - amin
                        dist = np.stack([np.abs(d1), np.abs(d2), np.abs(d3)])
                        exact0 = np.stack([d1 == 0, d2 == 0, d3 == 0])
                        a_box[bi] = (np.where(exact0, np.inf, dist).min(axis=0) <= 2 * e)
                dec_l = (amaxl >= aminl) & (amaxl > 0) & (aminl < bb)
                dec_r = (amaxr >= aminr) & (amaxr > 0) & (aminr < bb)
                any_cheap = a_box[0] | a_box[1] | (dec_l & dec_r & c4)
                any_sc = a_box_sc[0] | a_box_sc[1] | (sure_hit[0] & sure_hit[1] & c4)
                # a decision taken by the filter must be the exact one (the bound is rigorous): check it
                ok = ~any_cheap
                assert m == "ideal" or (np.array_equal(dec_l[ok], hl[ok]) and np.array_equal(dec_r[ok], hr[ok])), m
                if m in ("ray", "pair", "plane"):
                    assert np.array_equal((aminl > aminr)[ok & hl & hr], (tl > tr)[ok & hl & hr]), m
                elif m != "ideal":
                    sw = np.where(dom, False, aminl > aminr)
                    assert np.array_equal(sw[ok & hl & hr], (tl > tr)[ok & hl & hr]), m
                if m == "ideal":
                    any_cheap = a_box[0] | a_box[1] | (dec_l & dec_r & c4)
                if m == "pair+dom+flat":
                    direct[idx, ln[idx]] = flat_pair
                    any_cheap = any_cheap | flat_pair
                amb[m][idx, ln[idx]] = any_cheap
                amb_sc[m][idx, ln[idx]] = any_sc
            swap = hr & (~hl | (tl > tr))
            near = np.where(swap, left + 1, left)
            far = np.where(swap, left, left + 1)
            go = hl | hr
            b_idx = idx[both]
            stack[b_idx, sp[b_idx]] = far[both]
            sp[b_idx] += 1
            cur[idx[go]] = near[go]
            pop[idx[~go]] = True
        ln[live] += 1
        p = np.flatnonzero(pop)
        done = p[sp[p] == 0]
        live[done] = False
        cont = p[sp[p] > 0]
        sp[cont] -= 1
        cur[cont] = stack[cont, sp[cont]]
    return kind, ln, amb, amb_sc, ties, both_hits, direct


def replay(kind, amb, ln, valid, R=32, T=16, F=16, CF=None, CI=75.0, CL=70.0, CR=120.0, K_exact=0, direct=None):
    """The streamed walk: a wave deals spans of R x 64 rays to its lanes; every T trips, if >= F lanes are idle and rays are left, a
    refill look (CR).  One body per trip, the one most lanes are ready for.  CF None: today's two bodies.  K_exact > 0: the exact body
    only once that many lanes wait for it (or nothing else is ready)."""
    n = len(ln)
    cost = {"filter": 0.0, "inner": 0.0, "leaf": 0.0, "refill": 0.0}
    lanes = {"filter": 0, "inner": 0, "leaf": 0}
    steps = {"filter": 0, "inner": 0, "leaf": 0}
    span = R * 64
    for base in range(0, n, span):
        idx = [i for i in range(base, min(base + span, n)) if valid[i] and ln[i] > 0]
        nxt = 0
        ray = np.full(64, -1)
        pos = np.zeros(64, np.int64)
        parked = np.zeros(64, bool)          # filter ran, comparison undecided: waits for the exact body on the same node
        trips = 0
        while True:
            idle = ray < 0
            if nxt < len(idx) and ((trips % T == 0 and idle.sum() >= F) or idle.all()):
                for l in np.flatnonzero(idle):
                    if nxt < len(idx):
                        ray[l] = idx[nxt]; pos[l] = 0; parked[l] = False; nxt += 1
                cost["refill"] += CR
            act = ray >= 0
            if not act.any():
                break
            r = np.maximum(ray, 0)
            k = np.where(act, kind[r, pos], 255)
            at_leaf = k == 1
            at_inner = k == 0
            if CF is None:
                cand = {"inner": at_inner, "leaf": at_leaf}
            else:
                if direct is not None:
                    parked = parked | (at_inner & direct[r, pos])          # marked at upload: straight to the exact body
                cand = {"filter": at_inner & ~parked, "inner": at_inner & parked, "leaf": at_leaf}
            cnts = {b: int(m.sum()) for b, m in cand.items()}
            if CF is not None and K_exact and 0 < cnts["inner"] < K_exact and (cnts["filter"] or cnts["leaf"]):
                cnts["inner"] = 0
            # most lanes; ties: inner / filter before leaf (the kernel's rule: leaf only if strictly more)
            order = ("filter", "inner", "leaf") if CF is not None else ("inner", "leaf")
            body = max(order, key=lambda b: (cnts[b], -order.index(b)))
            m = cand[body]
            steps[body] += 1
            lanes[body] += int(m.sum())
            if body == "filter":
                cost["filter"] += CF
                a = amb[r, pos] & m
                parked |= a
                adv = m & ~a
            else:
                cost[body] += CI if body == "inner" else CL
                adv = m
                parked &= ~m
            pos[adv] += 1
            fin = adv & (pos >= ln[r])
            ray[fin] = -1
            trips += 1
    return cost, lanes, steps


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "DarkCornell"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    orc = Oracle()
    w = rpt.World.from_path(rpt.fixture(scene + ".glb"))
    sc = orc.scene(w)
    cfg = rpt.default_config(size, size)
    seeds = rpt.blue_noise_seeds(size, size)
    nodes_f = w.nodes.view(np.float32).reshape(-1, 8)
    nodes_u = w.nodes.view(np.uint32).reshape(-1, 8)
    verts = w.per_vertex["vertex"][:, :3].astype(np.float32)
    tris = w.indices.view(np.uint32).reshape(-1, 4)
    pmax = float(np.abs(nodes_f[1:, [0, 1, 2, 4, 5, 6]]).max())          # (the root's own box is never tested)
    order = hip.tile_order(size, size, 0, 1)
    px = (order >> 16).astype(np.int64) * size + (order & 0xFFFF).astype(np.int64)

    def tri_test(ti, o, d):                       # muller_trumbore (intersection.rs:9-54) in float32
        a, b, c = verts[tris[ti, 0]], verts[tris[ti, 1]], verts[tris[ti, 2]]
        e1, e2 = b - a, c - a
        pv = np.cross(d, e2).astype(np.float32)
        det = np.float32(np.dot(e1, pv))
        if abs(det) < 1e-6:
            return None
        inv = np.float32(1.0) / det
        tv = o - a
        u = np.float32(np.dot(tv, pv)) * inv
        if u < 0 or u > 1:
            return None
        qv = np.cross(tv, e1).astype(np.float32)
        v = np.float32(np.dot(d, qv)) * inv
        if v < 0 or u + v > 1:
            return None
        t = np.float32(np.dot(e2, qv)) * inv
        return t if t >= 0 else None

    print(f"{scene} {size}x{size}: largest |coordinate| of a tested box {pmax:.2f}")
    total = {}
    for bounce in range(4):
        rays = np.zeros((size * size, 6), np.float32)
        valid = np.zeros(size * size, np.uint8)
        orc.lib.oracle_dump_rays(C.byref(cfg), C.byref(sc), seeds.ctypes.data_as(C.c_void_p), C.c_uint32(bounce),
                                 rays.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
        rays, valid = rays[px], valid[px]
        o = np.where(valid[:, None] == 1, rays[:, :3], np.float32(0.0))
        d = np.where(valid[:, None] == 1, rays[:, 3:], np.float32(1.0))
        kind, ln, amb, amb_sc, ties, both, direct = walk(nodes_f, nodes_u, tri_test, o, d, pmax)
        ln = np.where(valid == 1, ln, 0)
        nr = int(valid.sum())
        inner_visits = sum(int((kind[i, :ln[i]] == 0).sum()) for i in range(len(ln)))
        leaf_visits = int(ln.sum()) - inner_visits
        line = f"bounce {bounce}: {nr} rays, {inner_visits / nr:.1f} box-pair + {leaf_visits / nr:.1f} leaf visits per ray; both children hit in {both / max(inner_visits, 1):.1%} of the visits, exact ties tl == tr in {ties / max(both, 1):.1%} of those; undecided visits"
        for m in MODES:
            a = sum(int(amb[m][i, :ln[i]].sum()) for i in range(len(ln)))
            s = sum(int(amb_sc[m][i, :ln[i]].sum()) for i in range(len(ln)))
            line += f"  [{m}] {a / inner_visits:.2%} (short-circuit form {s / inner_visits:.2%})"
        print(line)
        base_cost, base_lanes, base_steps = replay(kind, None, ln, valid)
        tb = sum(base_cost.values())
        print(f"    today        : {tb / nr:7.1f} VALU wave-instr/ray/64  inner lanes {base_lanes['inner'] / max(base_steps['inner'], 1):4.1f} leaf lanes {base_lanes['leaf'] / max(base_steps['leaf'], 1):4.1f}")
        total.setdefault("today", 0.0)
        total["today"] += tb
        total.setdefault("rays", 0)
        total["rays"] += nr
        for m in ("ray", "pair+dom", "pair+dom+flat", "ideal"):
            for CF in (48.0, 52.0, 56.0):
                for K in (0, 8):
                    if K and m != "pair+dom":
                        continue
                    c, l, s = replay(kind, amb[m], ln, valid, CF=CF + (2.0 if "dom" in m else 0.0), K_exact=K, direct=direct if m.endswith("flat") else None)
                    t = sum(c.values())
                    key = (m, CF, K)
                    total[key] = total.get(key, 0.0) + t
                    if CF == 52.0:
                        print(f"    {m:14s} CF {CF:.0f} K {K}: {t / nr:7.1f} ({t / tb - 1:+.1%})  filter lanes {l['filter'] / max(s['filter'], 1):4.1f} x {s['filter']}  exact lanes {l['inner'] / max(s['inner'], 1):4.1f} x {s['inner']}  leaf lanes {l['leaf'] / max(s['leaf'], 1):4.1f} x {s['leaf']}")
    print("all bounces, VALU wave-instructions of the walk relative to today's two-body loop:")
    for key, t in total.items():
        if isinstance(key, tuple):
            print(f"    bound per {key[0]:14s} filter body {key[1]:.0f} VALU, exact body once {key[2] or 1} lanes wait: {t / total['today'] - 1:+.1%}")


if __name__ == "__main__":
    main()
