"""Pin the CPU oracle to every known answer the reference holds for this path:
  * the furnace test, both NEE modes, at the reference's own settings   tests/correctness_tests.rs:14-53
  * the LDS sequence and blue-noise seed integer KATs                   kernels/src/rng.rs:20-32, src/trace.rs:150-157
  * BVH invariants + BVH == brute force                                  src/bvh.rs, kernels/src/intersection.rs:77-101
  * alias-table invariants                                                src/light_pick.rs:90-119
"""
import numpy as np
import pytest

from conftest import rel_l2


def test_lds_known_answers(oracle):
    # (n, dimension, offset) -> (u32 product, f32 value)   [SURVEY.md Appendix B.3]
    kats = [((1, 1, 0), 3144134276, 0.7320508360862732), ((2, 2, 0), 2027808484, 0.4721359610557556),
            ((0, 1, 1448498816), 2161089024, 0.5031677484512329), ((5, 3, 50529028), 1247431169, 0.2904402017593384),
            ((31, 2, 4294967295), 352356188, 0.08203931897878647)]
    for args, prod, val in kats:
        p, v = oracle.lds(*args)
        assert p == prod and v == val
    # u32 -> f32 is round-to-nearest-even: products >= 4294967168 give exactly 1.0
    assert np.float32(4294967168) * np.float32(1.0 / 4294967296.0) == np.float32(1.0)
    assert np.float32(4294967167) * np.float32(1.0 / 4294967296.0) == np.float32(0.99999994)


def test_blue_noise_seed_known_answers(rpt):
    tile = rpt.host.blue_noise_tile()
    assert tile.shape == (256, 256)
    assert list(tile[0, :8]) == [86, 100, 134, 40, 238, 185, 48, 249] and tile[75, 65] == 3
    seeds = rpt.blue_noise_seeds(300, 260)
    assert list(seeds["offset"][:4]) == [1448498816, 1684300928, 2256963328, 673720384]
    assert seeds["offset"][75 * 300 + 65] == 50529028 and np.all(seeds["n"] == 0)
    # wraps modulo the 256x256 tile (x % w, y % h)
    assert seeds["offset"][0] == seeds["offset"][256] == seeds["offset"][256 * 300]


@pytest.mark.parametrize("use_mis", [False, True])
def test_furnace_known_answer_cpu(oracle, rpt, world, use_mis):
    """furnace_test(use_cpu = true, use_mis): 128x128, 32 spp, pixel (65, 75), mean^(1/2.2) = 0.8 +- 0.02."""
    cfg = rpt.default_config(128, 128, nee=1 if use_mis else 0)
    acc, _, st = oracle.trace_cpu(cfg, oracle.scene(world("FurnaceTest")), rpt.blue_noise_seeds(128, 128), 32)
    assert st.error_flags == 0
    frame = acc[..., :3] / 32.0
    px = frame[75, 65] ** (1.0 / 2.2)
    assert np.all(np.abs(px - 0.8) < 0.02), px
    # stronger, converged form of the same statement: the disc of the inner sphere is a furnace
    yy, xx = np.mgrid[0:128, 0:128]
    disc = (xx - 64) ** 2 + (yy - 75) ** 2 < 8 ** 2
    g = frame[disc].mean(axis=0) ** (1.0 / 2.2)
    assert np.all(np.abs(g - 0.8) < 0.015), g


@pytest.mark.parametrize("nee", [0, 1])
def test_furnace_known_answer_at_baseline_config_0(oracle, rpt, world, nee):
    """BASELINE config[0] as stated: FurnaceTest.glb 256 x 256, 16 spp on the CPU path.  The reference's furnace assertion
    (tests/correctness_tests.rs:26-31) scaled to that size: the inner sphere's disc (centre (128, 150), radius 16 pixels) is 0.8 in
    gamma space within +- 0.02 in both estimator modes (oracle: 0.8075 without NEE, 0.8000 with MIS); at 16 spp a single pixel is
    inside the tolerance only without NEE (pixel (130, 150): 0.789), with MIS its spread is larger than 0.02."""
    cfg = rpt.default_config(256, 256, nee=nee)
    acc, _, st = oracle.trace_cpu(cfg, oracle.scene(world("FurnaceTest")), rpt.blue_noise_seeds(256, 256), 16)
    assert st.error_flags == 0 and np.all(acc[..., 3] == 16) and np.isfinite(acc).all()
    assert (st.extension_rays, st.shadow_rays) == ((1085345, 0), (1085346, 30470))[nee]
    frame = acc[..., :3] / np.float32(16)
    yy, xx = np.mgrid[0:256, 0:256]
    disc = (xx - 128) ** 2 + (yy - 150) ** 2 < 16 ** 2
    g = frame[disc].mean(axis=0).astype(np.float64) ** (1 / 2.2)
    assert np.all(np.abs(g - 0.8) < 0.02), g
    if nee == 0:
        px = frame[150, 130].astype(np.float64) ** (1 / 2.2)
        assert np.all(np.abs(px - 0.8) < 0.02), px


def test_furnace_known_answer_converged(oracle, rpt, world):
    """The reference's furnace KAT (tests/correctness_tests.rs:14-53) in CONVERGED form (at its own 32 spp
    the single pixel (65, 75) passes or fails by realisation).  512 spp, no NEE and MIS (the two modes the reference
    tests): pixel (65, 75) and the mean over the sphere's disc are 0.8 in gamma space within the reference's +-0.02; every
    pixel of the 5x5 neighbourhood is (+-0.02 without NEE, +-0.06 with MIS, whose per-pixel variance is higher).  The two
    estimators have the same expectation, so their converged disc means must agree: 0.8081 vs 0.8084 — and 0.808 is what
    SURVEY.md Appendix B.4 obtained from an independent f64 emulation of the reference's algorithm.
    Direct-only (nee = 2) is NOT a furnace in the reference: a diffuse bounce that reaches the emitter shades it as a
    surface (kernels/src/lib.rs:97-108 fall through), so the enclosure reflects as well as emits; only recorded here.
    This, the integer KATs and the struct layouts are everything the reference holds for this path; bit-level equality
    of the oracle with the Rust build itself cannot be checked in this image (no rustc)."""
    spp = 512
    rect = (44, 55, 86, 97)
    yy, xx = np.mgrid[0:128, 0:128]
    disc = (xx - 64) ** 2 + (yy - 75) ** 2 < 8 ** 2
    means = {}
    for nee in (0, 1, 2):
        cfg = rpt.default_config(128, 128, nee=nee)
        acc, _, st = oracle.trace_cpu(cfg, oracle.scene(world("FurnaceTest")), rpt.blue_noise_seeds(128, 128), spp, rect=rect)
        assert st.error_flags == 0 and np.isfinite(acc).all()
        frame = acc[..., :3] / np.float32(spp)
        g = np.power(frame, 1.0 / 2.2)
        means[nee] = np.power(frame[disc].mean(axis=0), 1.0 / 2.2)
        if nee == 2:
            continue
        assert np.all(np.abs(g[75, 65] - 0.8) < 0.02), g[75, 65]
        assert np.all(np.abs(g[73:78, 63:68] - 0.8) < (0.02, 0.06)[nee]), g[73:78, 63:68, 0]
        assert np.all(np.abs(means[nee] - 0.8) < 0.02) and np.all(np.abs(means[nee] - 0.808) < 0.003), means[nee]
    assert np.all(np.abs(means[0] - means[1]) < 0.002), means
    assert np.all(means[2] > 1.0)                       # the reference's direct-only mode gains energy in a closed emitter


def test_scene_statistics_match_survey(world):
    # triangles, materials (incl. assimp-style default for PBRTest), emissive triangles  [SURVEY.md Appendix B.1]
    expect = {"DarkCornell": (184, 8, 2), "VeachMIS": (2932, 6, 2880), "FurnaceTest": (10240, 2, 5120),
              "PBRTest": (24002, 26, 0)}
    for name, (tris, mats, emissive) in expect.items():
        w = world(name)
        assert (len(w.indices), len(w.materials), w.n_emissive_triangles) == (tris, mats, emissive)
        if emissive == 0:
            assert len(w.light_pick) == 1 and w.light_pick["ratio"][0] < 0
        else:
            assert len(w.light_pick) == emissive
    # DarkCornell light: emissiveFactor 0.6266 * 15 (src/asset.rs:165-168)
    em = world("DarkCornell").materials["emissive"]
    assert np.isclose(em[:, :3].max(), 0.6266 * 15, rtol=1e-3) and np.all(em[:, 3] == 15.0)   # alpha 1 * 15


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest", "PBRTest"])
def test_bvh_invariants(world, scene):
    w = world(scene)
    nodes, tris = w.nodes, w.indices
    n_tri = len(tris)
    assert len(nodes) <= 2 * n_tri - 1
    pos = w.per_vertex["vertex"][:, :3]
    covered = np.zeros(n_tri, np.int32)
    stack = [(0, 0)]
    visited = 0
    max_depth = 0
    while stack:
        i, d = stack.pop()
        visited += 1
        max_depth = max(max_depth, d)
        n = nodes[i]
        if n["triangle_count"] > 0:
            lo, cnt = int(n["left_or_first"]), int(n["triangle_count"])
            covered[lo:lo + cnt] += 1
            v = pos[np.stack([tris["v0"][lo:lo + cnt], tris["v1"][lo:lo + cnt], tris["v2"][lo:lo + cnt]], 1)].reshape(-1, 3)
            assert np.all(v >= n["aabb_min"] - 0) and np.all(v <= n["aabb_max"] + 0)
        else:
            l = int(n["left_or_first"])
            assert l + 1 < len(nodes)
            for c in (l, l + 1):   # children inside the parent box
                assert np.all(nodes[c]["aabb_min"] >= n["aabb_min"]) and np.all(nodes[c]["aabb_max"] <= n["aabb_max"])
            stack += [(l, d + 1), (l + 1, d + 1)]
    assert visited == len(nodes) and np.all(covered == 1)
    assert max_depth == w.bvh_max_depth <= 31


@pytest.mark.parametrize("scene,n_rays", [("DarkCornell", 20000), ("VeachMIS", 4000), ("PBRTest", 600)])
def test_bvh_traversal_equals_brute_force(oracle, world, scene, n_rays):
    w = world(scene)
    sc = oracle.scene(w)
    rng = np.random.default_rng(9)
    lo, hi = w.per_vertex["vertex"][:, :3].min(0), w.per_vertex["vertex"][:, :3].max(0)
    o = (lo + rng.random((n_rays, 3)) * (hi - lo)).astype(np.float32)
    d = rng.normal(size=(n_rays, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    t_b, tri_b, fl_b, e1 = oracle.trace_rays(sc, 0, o, d)
    t_s, tri_s, fl_s, e2 = oracle.trace_rays(sc, 2, o, d)
    assert e1 == 0 and e2 == 0
    assert np.array_equal(fl_b & 1, fl_s & 1)
    assert np.array_equal(t_b.view(np.uint32), t_s.view(np.uint32))      # same nearest t, bit for bit
    # any-hit is consistent with nearest: occluded within max_t  <=>  nearest t <= max_t
    max_t = (rng.random(n_rays) * 6).astype(np.float32)
    _, _, fl_a, _ = oracle.trace_rays(sc, 1, o, d, max_t)
    assert np.array_equal((fl_a & 1) == 1, ((fl_b & 1) == 1) & (t_b <= max_t))


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest"])
def test_alias_table_reproduces_power_pdf(world, scene):
    w = world(scene)
    lp = w.light_pick
    n = len(lp)
    mass = np.zeros(len(w.indices), np.float64)
    np.add.at(mass, lp["triangle_index_a"], lp["ratio"].astype(np.float64) / n)
    np.add.at(mass, lp["triangle_index_b"], (1.0 - lp["ratio"].astype(np.float64)) / n)
    pdf = np.zeros(len(w.indices), np.float64)
    pdf[lp["triangle_index_a"]] = lp["triangle_pick_pdf_a"]
    assert abs(pdf.sum() - 1.0) < 1e-3 and abs(mass.sum() - 1.0) < 1e-6
    assert np.all((lp["ratio"] >= 0) & (lp["ratio"] <= 1)) and np.all(lp["triangle_area_a"] > 0)
    # The reference's robin-hood fill (src/light_pick.rs:90-105) only tops up the bins BELOW the average and never
    # shrinks the donors' own bins to 1/n, so sampled mass tracks the power pdf only approximately — by design of
    # the reference, restated as is.  What does hold: topped-up bins are exactly full (p_a + p_b = average) ...
    avg = 1.0 / n
    topped = lp["ratio"] < 1.0
    if topped.any():
        p_a = lp["triangle_pick_pdf_a"][topped].astype(np.float64)
        full = p_a / lp["ratio"][topped].astype(np.float64)
        assert np.allclose(full, avg, rtol=2e-3)
    # ... and the sampled mass stays within a factor of the pdf everywhere
    assert np.abs(mass - pdf).sum() < 0.4


def test_rng_state_advances_and_resume(oracle, rpt, world):
    cfg = rpt.default_config(48, 32)
    sc = oracle.scene(world("DarkCornell"))
    seeds = rpt.blue_noise_seeds(48, 32)
    a, rng_a, _ = oracle.trace_cpu(cfg, sc, seeds, 5, threads=3)
    assert np.all(rng_a["n"] == 5) and np.array_equal(rng_a["offset"], seeds["offset"])
    b1, rng_b, _ = oracle.trace_cpu(cfg, sc, seeds, 2, threads=1)
    b2, _, _ = oracle.trace_cpu(cfg, sc, rng_b, 3, accum=b1, threads=8)
    assert np.array_equal(a.view(np.uint32), b2.view(np.uint32))      # thread count / batching never changes the sum


def test_oracle_with_platform_libm_stays_close(oracle, oracle_libm, rpt, world):
    """How far is the shared deterministic math from 'what glibc gives' at image level."""
    for scene, nee in [("DarkCornell", 1), ("VeachMIS", 1)]:
        cfg = rpt.default_config(64, 64, nee=nee)
        seeds = rpt.blue_noise_seeds(64, 64)
        a, _, _ = oracle.trace_cpu(cfg, oracle.scene(world(scene)), seeds, 8)
        b, _, _ = oracle_libm.trace_cpu(cfg, oracle_libm.scene(world(scene)), seeds, 8)
        err = rel_l2(a[..., :3], b[..., :3])
        frac = float((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).mean())
        print(f"{scene}: libm vs rpt_math rel-L2 {err:.2e}, pixels differing {frac:.3%}")
        assert err < 5e-2     # a flipped lobe/roulette decision changes single samples; energy stays put
        assert abs(a[..., :3].mean() - b[..., :3].mean()) / b[..., :3].mean() < 5e-3


def test_float_only_sky_exp_stays_within_1e_6_of_libm(oracle, oracle_libm, rpt, world):
    """The one place where the oracle follows the kernel instead of a correctly rounded function: the sky march's 84 exp per
    miss go through rpt_math.h `exp_sky` (<= 1 ulp, float only) on both sides.  No path decision reads sky radiance, so the
    effect must be a relative difference of the radiance itself — measured here against the libm build (glibc expf, also
    <= 1 ulp), where it is isolated from flipped decisions: (a) skybox::scatter alone on 20 000 directions, (b) the
    sky-dominated BASELINE scenes with max_bounces = 1 (a miss adds the sky, a hit adds emission or nothing: no lobe choice,
    no light pick, no roulette can differ).  Bound: 1e-6 rel-L2, a hundredth of the 1e-4 contract."""
    cfg0 = rpt.default_config(64, 64)
    sun = np.array(list(cfg0.sun_direction), np.float32)
    rng = np.random.default_rng(11)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d[:, 1] = np.abs(d[:, 1]) * rng.choice([1.0, 0.02], len(d)).astype(np.float32)        # upper hemisphere, many near the horizon
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = np.array(list(cfg0.cam_position)[:3], np.float32)
    a, b = oracle.sky(sun, org, d), oracle_libm.sky(sun, org, d)
    ok = np.isfinite(a).all(axis=1) & np.isfinite(b).all(axis=1)
    assert ok.mean() > 0.99
    err = rel_l2(a[ok], b[ok])
    worst = float(np.max(np.abs(a[ok].astype(np.float64) - b[ok]) / np.maximum(np.abs(b[ok]), 1e-20)))
    print(f"sky alone: rel-L2 {err:.2e}, worst component {worst:.2e}, identical {np.mean((a[ok] == b[ok]).all(axis=1)):.1%}")
    assert err <= 1e-6 and worst <= 1e-5
    for scene in ("VeachMIS", "PBRTest"):
        cfg = rpt.default_config(96, 64, max_bounces=1, min_bounces=0)
        seeds = rpt.blue_noise_seeds(96, 64)
        ia, _, sa = oracle.trace_cpu(cfg, oracle.scene(world(scene)), seeds, 4)
        ib, _, sb = oracle_libm.trace_cpu(cfg, oracle_libm.scene(world(scene)), seeds, 4)
        assert sa.sky_evals == sb.sky_evals and sa.sky_evals > 1000                         # same misses; the rest of the image is emission or black
        e = rel_l2(ia[..., :3], ib[..., :3])
        print(f"{scene} (primary rays only): rel-L2 {e:.2e}, {sa.sky_evals} sky evaluations")
        assert e <= 1e-6
