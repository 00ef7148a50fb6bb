"""The any-hit (shadow) walk may visit siblings in ANY order: the reference's shadow query reads `.hit` only (kernels/src/light_pick.rs:148),
`result.t` stays 1e6 until the first accept, which returns (intersection.rs:191-203), and boxes are pruned against that constant (:212-213) —
so `.hit` is the OR of the accept test over the triangles of the leaves whose ancestors' boxes the ray hits, whatever the order.  Proven here on
10^6 random shadow rays through the oracle's own box and triangle tests, and the upload-time choice the product derives from it (csrc/shadow_order.h)
is pinned: which scenes flip, that it is deterministic, that a flipped tree is the same tree."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_ffi import _p  # noqa: E402


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    """tools/anyhit_order_sim.cpp (includes the oracle's translation unit: its intersect_aabb / muller_trumbore / intersect_front_to_back)"""
    so = tmp_path_factory.mktemp("anyhit") / "libanyhit_sim.so"
    subprocess.run(["g++", "-std=c++20", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-msse4.1", "-pthread", "-shared", "-o", str(so),
                    os.path.join(ROOT, "tools", "anyhit_order_sim.cpp")], check=True)
    return C.CDLL(str(so))


def _shadow_like_rays(rng, n, world):
    """origins inside the scene's bounds, directions towards random points of it (so that many rays end ON geometry), max_t around that distance —
    plus rays with exact zeros in the direction (infinite slab distances) and rays that start on a vertex"""
    v = world.per_vertex["vertex"][:, :3]
    lo, hi = v.min(axis=0), v.max(axis=0)
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    target = v[rng.integers(0, len(v), n)] + (rng.normal(size=(n, 3)) * 0.05 * (hi - lo)).astype(np.float32)
    d = (target - o).astype(np.float32)
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    d /= np.maximum(dist, 1e-6)[:, None]
    d[:2000, 0] = 0.0
    d[2000:4000, 1] = 0.0
    d[4000:5000] = np.array([0.0, 0.0, 1.0], np.float32)
    o[5000:7000] = v[rng.integers(0, len(v), 2000)]
    max_t = (dist * rng.choice([0.5, 0.999, 1.0, 1.001, 2.0, 100.0], n)).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d.astype(np.float32)), np.ascontiguousarray(max_t)


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest", "PBRTest"])
def test_any_visiting_order_gives_the_references_hit(sim, oracle, rpt, world, scene):
    w = world(scene)
    sc = oracle.scene(w)
    rng = np.random.default_rng(41)
    n = 250_000
    o, d, max_t = _shadow_like_rays(rng, n, w)
    ref = np.zeros(n, np.uint8)
    sim.sim_any_hit_order(C.byref(sc), C.c_size_t(n), _p(o), _p(d), _p(max_t), 0, C.c_uint32(0), _p(ref))       # intersect_front_to_back<false> itself
    assert 0.02 * n < ref.sum() < 0.98 * n                   # both outcomes are well represented
    for mode, seed in ((1, 0), (2, 0), (3, 0), (4, 1), (4, 2), (5, 0)):   # left / right / far first, random per (ray, node) twice, breadth-first
        got = np.zeros(n, np.uint8)
        sim.sim_any_hit_order(C.byref(sc), C.c_size_t(n), _p(o), _p(d), _p(max_t), mode, C.c_uint32(seed), _p(got))
        assert np.array_equal(got, ref), (scene, mode, int((got != ref).sum()))


def test_the_upload_time_choice_of_the_shadow_order(hipmod, rpt, world, monkeypatch):
    """csrc/shadow_order.h through rpt_debug_shadow_order_host (host code, no GPU): DarkCornell's shadow rays find their occluder sooner opaque-first,
    FurnaceTest's walk the same nodes either way, a scene without lights is not probed; the choice is deterministic and can be overridden."""
    monkeypatch.delenv("RPT_SHADOW_ORDER", raising=False)
    dc = hipmod.shadow_order_host(world("DarkCornell"))
    assert dc["fixed"] and dc["probe_rays"] > 1500 and dc["visits_fixed"] < 0.7 * dc["visits_near"]      # (of 4 096 candidates: the ones that decide something)
    assert 0 < dc["flip"].sum() < len(dc["flip"])
    again = hipmod.shadow_order_host(world("DarkCornell"))
    assert again["visits_near"] == dc["visits_near"] and again["visits_fixed"] == dc["visits_fixed"] and np.array_equal(again["flip"], dc["flip"])
    vm = hipmod.shadow_order_host(world("VeachMIS"))
    assert vm["fixed"] == (vm["visits_fixed"] < 0.95 * vm["visits_near"]) and vm["probe_rays"] > 1000
    ft = hipmod.shadow_order_host(world("FurnaceTest"))     # an emitter all around a sphere: nothing but the sphere itself can occlude, every order walks the same nodes
    assert not ft["fixed"] and abs(ft["visits_fixed"] - ft["visits_near"]) < 0.02 * ft["visits_near"]
    pb = hipmod.shadow_order_host(world("PBRTest"))
    assert not pb["fixed"] and pb["probe_rays"] == 0         # sentinel light table: no shadow rays, nothing to choose
    monkeypatch.setenv("RPT_SHADOW_ORDER", "fixed")
    assert hipmod.shadow_order_host(world("VeachMIS"))["fixed"]
    monkeypatch.setenv("RPT_SHADOW_ORDER", "near")
    assert not hipmod.shadow_order_host(world("DarkCornell"))["fixed"]


def test_a_flipped_tree_is_the_same_tree(sim, hipmod, oracle, rpt, world):
    """What the fixed-order kernels walk: the node pool with the two nodes of every flipped pair exchanged.  Same leaves, same boxes, same `.hit`
    under the reference's own walk."""
    import copy
    w = world("DarkCornell")
    flip = hipmod.shadow_order_host(w)["flip"]
    w2 = copy.copy(w)
    nodes = w.nodes.copy()
    for p in np.nonzero(flip)[0]:
        nodes[[2 * p + 1, 2 * p + 2]] = nodes[[2 * p + 2, 2 * p + 1]]
    w2.nodes = nodes
    rng = np.random.default_rng(43)
    n = 100_000
    o, d, max_t = _shadow_like_rays(rng, n, w)
    a, b = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    sc, sc2 = oracle.scene(w), oracle.scene(w2)
    sim.sim_any_hit_order(C.byref(sc), C.c_size_t(n), _p(o), _p(d), _p(max_t), 0, C.c_uint32(0), _p(a))
    sim.sim_any_hit_order(C.byref(sc2), C.c_size_t(n), _p(o), _p(d), _p(max_t), 1, C.c_uint32(0), _p(b))          # left first = the preferred child first
    assert np.array_equal(a, b) and 0 < a.sum() < n


@pytest.mark.parametrize("scene", ["DarkCornell", "PBRTest"])
def test_the_nearest_walk_says_hit_at_its_first_accept_in_any_order(sim, oracle, rpt, world, scene):
    """What the last extension rays of a batch without NEE rest on (k_traverse.h k_traverse_nearest_stream LAST): intersect_nearest's `.hit`
    (intersection.rs:169-171) is decided at the first triangle its walk accepts — result.t is 1e6 until then, exactly the any-hit walk with
    max_t = 1e6 — so it equals the any-hit answer under EVERY visiting order; and a ray whose nearest hit is an emissive triangle passes that
    triangle's Moller-Trumbore test (so a ray that passes none cannot end on one)."""
    w = world(scene)
    sc = oracle.scene(w)
    rng = np.random.default_rng(47)
    n = 200_000
    o, d, _ = _shadow_like_rays(rng, n, w)
    t, tri, flags, _ = oracle.trace_rays(sc, 0, o, d)                           # intersect_front_to_back<true>
    hit = (flags & 1).astype(np.uint8)
    assert 0.02 * n < hit.sum() < 0.999 * n
    far = np.full(n, 1000000.0, np.float32)
    for mode, seed in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 1), (5, 0)):
        got = np.zeros(n, np.uint8)
        sim.sim_any_hit_order(C.byref(sc), C.c_size_t(n), _p(o), _p(d), _p(far), mode, C.c_uint32(seed), _p(got))
        assert np.array_equal(got, hit), (scene, mode, int((got != hit).sum()))
    em = w.materials["emissive"][:, :3]
    em_tris = np.nonzero(np.isin(w.indices["material"], [m for m in range(len(em)) if np.any(em[m] != 0)]))[0]
    if len(em_tris):
        # rays aimed at the emitters: the ones whose walk ends on one are accepted by that triangle in the brute-force walk too (mode 2 tests every
        # triangle on its own: the Moller-Trumbore test of the emitter passed)
        m = 40_000
        v = w.per_vertex["vertex"][:, :3]
        lo, hi = v.min(axis=0), v.max(axis=0)
        o2 = (lo + (hi - lo) * rng.random((m, 3))).astype(np.float32)
        tr = w.indices[em_tris[rng.integers(0, len(em_tris), m)]]
        bary = rng.dirichlet((1.0, 1.0, 1.0), m).astype(np.float32)
        target = bary[:, :1] * v[tr["v0"]] + bary[:, 1:2] * v[tr["v1"]] + bary[:, 2:] * v[tr["v2"]]
        d2 = (target - o2).astype(np.float32)
        d2 /= np.maximum(np.linalg.norm(d2, axis=1), 1e-6)[:, None].astype(np.float32)
        o2, d2 = np.ascontiguousarray(o2), np.ascontiguousarray(d2.astype(np.float32))
        _, tri2, flags2, _ = oracle.trace_rays(sc, 0, o2, d2)
        ends_on_emitter = ((flags2 & 1) == 1) & np.isin(tri2, em_tris)
        assert ends_on_emitter.sum() > 1000          # (the light is occluded from most of the box by its own housing)
        _, tri_b, flags_b, _ = oracle.trace_rays(sc, 2, o2[ends_on_emitter], d2[ends_on_emitter])
        assert np.all(flags_b & 1)


def test_the_upload_time_choice_of_the_last_rays_order(hipmod, rpt, world, monkeypatch):
    """csrc/shadow_order.h choose_last_order through rpt_debug_last_order_host (host code, no GPU): on the closed box every fixed rule finds a first hit
    in fewer node visits than the reference's near-first order and the best one is taken; deterministic; RPT_LAST_ORDER overrides; the flip it hands
    out is a proper subset of the pairs."""
    monkeypatch.delenv("RPT_LAST_ORDER", raising=False)
    dc = hipmod.last_order_host(world("DarkCornell"))
    assert dc["rule"] in (1, 2, 3) and dc["probe_rays"] > 500
    assert dc["visits"][dc["rule"]] == min(dc["visits"][1:]) < 0.95 * dc["visits"][0]
    assert 0 < dc["flip"].sum() < len(dc["flip"])
    again = hipmod.last_order_host(world("DarkCornell"))
    assert again["visits"] == dc["visits"] and np.array_equal(again["flip"], dc["flip"])
    for name, rule in (("near", 0), ("opaque", 1), ("small", 2), ("ratio", 3)):
        monkeypatch.setenv("RPT_LAST_ORDER", name)
        assert hipmod.last_order_host(world("DarkCornell"))["rule"] == rule
