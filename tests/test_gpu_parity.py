"""GPU parity tests: the HIP wavefront path (through the rpt.h C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): per-pixel relative L2 <= 1e-4 on the mean
radiance image.  Because rpt_math.h makes both sides bit-identical, the tests
assert the STRONGER property first (bitwise equal accumulators, equal ray
counts) and report rel-L2 as the contractual figure.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_REL_L2 = 1e-4   # north_star tolerance


def _random_rays(rng, n, world):
    lo = world.per_vertex["vertex"][:, :3].min(axis=0)
    hi = world.per_vertex["vertex"][:, :3].max(axis=0)
    ext = np.maximum(hi - lo, 1e-3)
    o = (lo - 0.3 * ext + rng.random((n, 3)) * 1.6 * ext).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    # a few axis-aligned directions: exercise the 0-component / inf / NaN slab paths
    d[: n // 50] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 50)] * rng.choice([-1.0, 1.0], (n // 50, 1)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("op,lo,hi", [(0, -10, 10), (1, -10, 10), (2, -1, 1), (3, -100, 30), (5, -1, 1), (7, 0, 1e6), (10, -110, 30)])
def test_math_bitwise_device_vs_host(renderer, hipmod, oracle, op, lo, hi):
    rng = np.random.default_rng(op)
    x = rng.uniform(lo, hi, 1 << 20).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.float32(hi), np.float32(lo)]
    dev = renderer.debug_math(op, x)
    host = hipmod.debug_math_host(op, x)          # clang host build inside librpt_hip.so
    orc = oracle.math(op, x)                      # g++ build inside the oracle
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))
    assert np.array_equal(dev.view(np.uint32), orc.view(np.uint32))


def test_math_two_operand_bitwise(renderer, oracle):
    rng = np.random.default_rng(11)
    n = 1 << 20
    x = rng.uniform(0, 4, n).astype(np.float32)
    y = rng.uniform(0.1, 3, n).astype(np.float32)
    y[: n // 2] = np.float32(2.2)
    assert np.array_equal(renderer.debug_math(4, x, y).view(np.uint32), oracle.math(4, x, y).view(np.uint32))   # pow
    a = rng.uniform(-3, 3, n).astype(np.float32)
    b = rng.uniform(-3, 3, n).astype(np.float32)
    assert np.array_equal(renderer.debug_math(6, a, b).view(np.uint32), oracle.math(6, a, b).view(np.uint32))   # atan2
    b[:100] = 0.0   # x / 0, 0 / 0
    a[:50] = 0.0
    q_dev, q_cpu = renderer.debug_math(8, a, b), oracle.math(8, a, b)                                          # IEEE divide
    both_nan = np.isnan(q_dev) & np.isnan(q_cpu)
    assert np.array_equal(q_dev.view(np.uint32)[~both_nan], q_cpu.view(np.uint32)[~both_nan])


def test_texel_channels_are_divided_by_255_exactly_on_the_device(renderer):
    """rptm::unorm8 == u8 / 255.0f for all 256 channel values (the textured shade stage converts up to 48 per hit)."""
    x = np.arange(256, dtype=np.float32)
    got = renderer.debug_math(11, x)
    assert np.array_equal(got.view(np.uint32), (x / np.float32(255.0)).view(np.uint32))


def test_fast_division_on_device_equals_ieee(renderer):
    """rpt_fastdiv.h on gfx950: guarded reciprocal division == IEEE division (modulo the sign of a zero quotient)."""
    rng = np.random.default_rng(23)
    n = 1 << 22
    y = rng.uniform(-1, 1, n).astype(np.float32)
    y[: n // 8] = (rng.uniform(-1, 1, n // 8) * 10.0 ** rng.uniform(-12, 0, n // 8)).astype(np.float32)
    x = (rng.uniform(-50, 50, n) - rng.uniform(-50, 50, n)).astype(np.float32)
    x[:64] = 0.0
    fast, true = renderer.debug_math(9, x, y), renderer.debug_math(8, x, y)
    ok = (np.isnan(fast) & np.isnan(true)) | ((fast == 0) & (true == 0))
    assert np.array_equal(fast.view(np.uint32)[~ok], true.view(np.uint32)[~ok])


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest", "PBRTest"])
def test_ray_parity_nearest_and_any(renderer, oracle, world, scene):
    w = world(scene)
    renderer.upload_scene(w)
    sc = oracle.scene(w)
    rng = np.random.default_rng(5)
    o, d = _random_rays(rng, 200_000, w)
    t_g, tri_g, fl_g = renderer.debug_trace_rays(False, o, d)
    t_c, tri_c, fl_c, err = oracle.trace_rays(sc, 0, o, d)
    assert err == 0
    assert np.array_equal(fl_g, fl_c)
    hit = (fl_c & 1) == 1
    assert hit.sum() > 1000
    assert np.array_equal(t_g.view(np.uint32), t_c.view(np.uint32))
    assert np.array_equal(tri_g[hit], tri_c[hit])
    max_t = (rng.random(len(o)) * 8).astype(np.float32)
    _, _, afl_g = renderer.debug_trace_rays(True, o, d, max_t)
    _, _, afl_c, err = oracle.trace_rays(sc, 1, o, d, max_t)
    assert err == 0
    assert np.array_equal(afl_g & 1, afl_c & 1)


@pytest.mark.parametrize("scene,knob", [("DarkCornell", None), ("DarkCornell", "RPT_NO_LDS_SCENE=1"),
                                        ("VeachMIS", None), ("VeachMIS", "foreign_pool"), ("FurnaceTest", None), ("PBRTest", None),
                                        ("PBRTest", "RPT_COOP_LEAVES=1"), ("deep_bvh", None), ("deep_bvh", "RPT_COOP_LEAVES=0"),
                                        ("deep_bvh", "RPT_STACK_BITS=21"), ("scatter", None)])
def test_ray_parity_through_the_production_traversal_stage(monkeypatch, hipmod, oracle, rpt, world, scene, knob):
    """intersect_front_to_back (intersection.rs:177-234) per ray — t, triangle, backface bit for bit against the oracle — through
    the kernels rpt_render itself launches (rpt_debug_trace_rays_production): the persistent LDS-pool stream for DarkCornell, the
    streamed global-memory walks for the others (without the cooperative leaf code for thin-leaf scenes, with it for the fat-leaf
    stand-in, and each forced the other way), the generic one-ray-per-lane walk for a node pool that is not pair-shaped (a foreign
    builder's: scenes.foreign_pool), 21-bit stack entries.  The rays are
    incoherent (random origins inside the scene, random directions) and include axis-parallel directions with exact zeros,
    which leave the exact-division fast path."""
    if knob and "=" in knob:
        k, v = knob.split("=")
        monkeypatch.setenv(k, v)
    if scene == "deep_bvh":
        from scenes import deep_bvh_scene
        w = deep_bvh_scene(60_000)
    elif scene == "scatter":
        from scenes import scatter_scene
        w = scatter_scene(80_000)                                # > 65 536 nodes: stack entries wider than 16 bits
    else:
        w = world(scene)
    if knob == "foreign_pool":
        from scenes import foreign_pool
        w = foreign_pool(w)
    n = 150_000
    rng = np.random.default_rng(17)
    o, d = _random_rays(rng, n, w)
    d[:300, 0] = 0.0                                             # a zero direction component: infinite slab distances
    d[300:600, 1] = 0.0
    d[600:700] = np.array([0.0, 0.0, 1.0], np.float32)
    r = hipmod.Renderer(0)
    r.upload_scene(w)
    r.set_config(rpt.default_config(512, 512))                   # 512 x 512 x slots per pixel >= n slots
    r.reset(rpt.blue_noise_seeds(512, 512))
    t_g, tri_g, fl_g = r.debug_trace_rays_production(o, d)
    t_c, tri_c, fl_c, err = oracle.trace_rays(oracle.scene(w), 0, o, d)
    assert err == 0
    hit = (fl_c & 1) == 1
    assert hit.sum() > 1000 and (~hit).sum() >= 0
    assert np.array_equal(fl_g, fl_c)
    assert np.array_equal(t_g.view(np.uint32), t_c.view(np.uint32))
    assert np.array_equal(tri_g[hit], tri_c[hit])
    # the context is usable afterwards (after a reset): one sample renders and equals the oracle's
    cfg = rpt.default_config(64, 48)
    r.set_config(cfg); r.reset(rpt.blue_noise_seeds(64, 48))
    r.render(1)
    acc, _ = r.read_accum()
    ref, _, _ = oracle.trace_cpu(cfg, oracle.scene(w), rpt.blue_noise_seeds(64, 48), 1)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    r.close()


CASES = [
    # scene, W, H, spp, nee, config overrides
    ("FurnaceTest", 128, 128, 8, 0, {}),
    ("FurnaceTest", 128, 128, 8, 1, {}),
    ("FurnaceTest", 96, 64, 4, 2, {}),
    ("FurnaceTest", 256, 256, 16, 0, {}),                      # BASELINE config[0] at its own size and sample count
    ("FurnaceTest", 256, 256, 16, 1, {}),
    ("DarkCornell", 256, 256, 8, 0, {}),
    ("DarkCornell", 200, 120, 6, 1, {}),                       # ragged: not a multiple of the 64-pixel tile
    ("DarkCornell", 128, 128, 4, 2, {"min_bounces": 1, "max_bounces": 3}),   # roulette active
    ("VeachMIS", 192, 108, 4, 1, {}),
    ("VeachMIS", 160, 90, 4, 0, {"cam_rotation": (0.2, -0.4, 0.0, 0.0), "cam_position": (1.0, 2.0, -6.0, 0.0)}),
    ("PBRTest", 128, 128, 2, 0, {}),
    ("PBRTest", 128, 96, 2, 1, {"min_bounces": 0, "max_bounces": 3}),
]


@pytest.mark.parametrize("scene,W,H,spp,nee,over", CASES)
def test_image_parity_with_oracle(renderer, oracle, rpt, world, scene, W, H, spp, nee, over):
    w = world(scene)
    cfg = rpt.default_config(W, H, nee=nee, **over)
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc_g, samples = renderer.read_accum()
    st_g = renderer.stats()
    acc_c, rng_c, st_c = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    assert samples == spp and st_c.error_flags == 0
    assert np.all(acc_g[..., 3] == spp)
    # ray accounting must agree exactly
    assert st_g["extension_rays"] == st_c.extension_rays
    assert st_g["shadow_rays"] == st_c.shadow_rays
    assert st_g["sky_evals"] == st_c.sky_evals
    if nee == 0:
        assert st_g["shadow_rays_elided"] == 0
    elif W <= 200:
        # NEE evaluations whose shadow ray decides nothing (light_pdf = 0 or bsdf_pdf = 0: the term is zero whatever the walk finds, light_pick.rs:150-158)
        # are not walked on the device: exactly the ones the oracle's analysis hook identifies, and their share is what tools/dead_shadow_rays.py reports
        n_all, n_dead, _, _ = oracle.dead_shadow_rays(cfg, oracle.scene(w), seeds, spp)
        assert n_all == st_c.shadow_rays and st_g["shadow_rays_elided"] == n_dead
        assert st_g["shadow_rays_traced"] == n_all - n_dead
        if st_c.shadow_rays > 10000:
            assert 0.2 * n_all < n_dead < 0.8 * n_all
    # rng[i].x += 1 per sample (kernels/src/lib.rs:226)
    rng_g = renderer.read_rng()
    assert np.array_equal(rng_g["n"], rng_c["n"]) and np.array_equal(rng_g["offset"], rng_c["offset"])
    mean_g, mean_c = acc_g[..., :3] / spp, acc_c[..., :3] / spp
    err = rel_l2(mean_g, mean_c)
    n_diff = int((acc_g.view(np.uint32) != acc_c.view(np.uint32)).any(axis=2).sum())
    print(f"{scene} {W}x{H} spp={spp} nee={nee}: relL2={err:.3e} pixels differing bitwise={n_diff}")
    assert err <= TOL_REL_L2
    assert n_diff == 0, "accumulators are expected to be bit-identical to the oracle"


@pytest.mark.parametrize("scene,W,H,nee,spp", [("DarkCornell", 1024, 1024, 1, 2), ("VeachMIS", 1920, 1080, 1, 1)])
def test_elided_shadow_rays_equal_the_oracles_count_at_full_size(hipmod, oracle, rpt, world, scene, W, H, nee, spp):
    """The NEE evaluations whose shadow ray decides nothing (k_shade.h; light_pick.rs:141-172 + lib.rs:164) at the BASELINE configurations' own resolutions:
    the device elides exactly the evaluations the oracle's analysis hook identifies, walks exactly the others, and the image is the oracle's, whole."""
    w = world(scene)
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
        r.render(spp)
        acc, _ = r.read_accum()
        st = r.stats()
    finally:
        r.close()
    sc = oracle.scene(w)
    ref, _, st_c = oracle.trace_cpu(cfg, sc, seeds, spp)
    n_all, n_dead, _, _ = oracle.dead_shadow_rays(cfg, sc, seeds, spp)
    assert n_all == st_c.shadow_rays == st["shadow_rays"] and st["shadow_rays_elided"] == n_dead and st["shadow_rays_traced"] == n_all - n_dead
    assert 0.2 * n_all < n_dead < 0.8 * n_all and st["extension_rays"] == st_c.extension_rays
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("order", ["near", "fixed"])
@pytest.mark.parametrize("scene,W,H,spp,nee,over", [c for c in CASES if c[4] != 0 and c[1] <= 200 and c[0] != "PBRTest"] + [("deep_bvh", 96, 64, 2, 1, {"cam_position": (0.0, 2.5, -0.5, 0.0)}),
                                                                                                  ("scatter", 96, 64, 2, 1, {"cam_position": (0.0, 1.8, -0.9, 0.0)})])
def test_either_shadow_order_gives_the_oracles_image(monkeypatch, hipmod, oracle, rpt, world, scene, W, H, spp, nee, over, order):
    """The shadow (any-hit) walks may visit siblings in any order (light_pick.rs:148 reads `.hit` only; tests/test_anyhit_order.py).  The library picks
    per scene at upload (rpt_shadow_order); forced either way — the reference's near-first order, or the fixed opaque-first order over the flipped
    copy of the tree (LDS image for DarkCornell, pair arrays for the others, thin and fat leaves, 16- and 21-bit stack entries) — every NEE image is
    the oracle's bit for bit, and so are the ray counts."""
    monkeypatch.setenv("RPT_SHADOW_ORDER", order)
    if scene == "deep_bvh":
        from scenes import deep_bvh_scene
        w = deep_bvh_scene(60_000)
    elif scene == "scatter":
        from scenes import scatter_scene
        w = scatter_scene(80_000)
    else:
        w = world(scene)
    cfg = rpt.default_config(W, H, nee=nee, **over)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w)
        so = r.shadow_order()
        assert so["fixed"] == (order == "fixed") and so["probe_rays"] > 0
        r.set_config(cfg); r.reset(seeds)
        r.render(spp)
        acc_g, _ = r.read_accum()
        st_g = r.stats()
    finally:
        r.close()
    acc_c, _, st_c = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    assert st_g["shadow_rays"] == st_c.shadow_rays > 0 and st_g["extension_rays"] == st_c.extension_rays
    assert np.array_equal(acc_g.view(np.uint32), acc_c.view(np.uint32))


def test_the_library_chooses_the_shadow_order_per_scene(monkeypatch, hipmod, rpt, world):
    """rpt_shadow_order after rpt_upload_scene: DarkCornell walks its shadow rays opaque-first (the probe rays find their occluders in half the node
    visits), VeachMIS whichever its probe favours, PBRTest (no lights) is not probed; re-uploading another scene into the same context re-decides."""
    monkeypatch.delenv("RPT_SHADOW_ORDER", raising=False)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(world("DarkCornell"))
        so = r.shadow_order()
        assert so["fixed"] and so["visits_fixed"] < 0.7 * so["visits_near"] and {k: v for k, v in so.items() if k != "probe_ms"} == {k: v for k, v in hipmod.shadow_order_host(world("DarkCornell")).items() if k != "flip"}
        r.upload_scene(world("VeachMIS"))
        vm = r.shadow_order()
        assert vm["fixed"] == (vm["visits_fixed"] < 0.95 * vm["visits_near"]) and vm["probe_rays"] > 1000 and vm["probe_ms"] > 0
        r.upload_scene(world("PBRTest"))
        assert not r.shadow_order()["fixed"] and r.shadow_order()["probe_rays"] == 0
        r.upload_scene(world("DarkCornell"))
        assert r.shadow_order()["fixed"]
    finally:
        r.close()


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest", "PBRTest", "deep_bvh", "scatter", "textured", "fat_leaf"])
def test_the_probe_kernels_decide_exactly_as_the_host_loop_does(monkeypatch, hipmod, rpt, world, scene):
    """Round 6: rpt_upload_scene runs both order probes as kernels (per-node sums level by level, one thread per probe ray); the sequential host loop over the
    same core (rpt_debug_shadow_order_host / rpt_debug_last_order_host) is the checker.  Same rays, same node visits — the averages are EQUAL, digit for digit,
    and so are the probe-ray counts and the decisions — on the four shipped scenes, the two 1 M-triangle stand-ins' smaller brothers (deep tree with fat
    leaves; more than 65 536 nodes), a textured open scene and a leaf of 300 triangles."""
    for k in ("RPT_SHADOW_ORDER", "RPT_LAST_ORDER"):
        monkeypatch.delenv(k, raising=False)
    if scene == "deep_bvh":
        from scenes import deep_bvh_scene
        w = deep_bvh_scene(120_000)
    elif scene == "scatter":
        from scenes import scatter_scene
        w = scatter_scene(70_000)
    elif scene == "textured":
        from scenes import textured_scene
        w = textured_scene()[0]
    elif scene == "fat_leaf":
        from scenes import fat_leaf_scene
        w = fat_leaf_scene(300)
    else:
        w = world(scene)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w)
        dev_s, dev_l = r.shadow_order(), r.last_bounce_order()
    finally:
        r.close()
    host_s, host_l = hipmod.shadow_order_host(w), hipmod.last_order_host(w)
    assert (dev_s["fixed"], dev_s["visits_near"], dev_s["visits_fixed"], dev_s["probe_rays"]) == (host_s["fixed"], host_s["visits_near"], host_s["visits_fixed"], host_s["probe_rays"])
    assert dev_s["probe_ms"] > 0
    if host_s["probe_rays"]:                                   # (no lights: no shadow probe; a leaf of 255+ triangles: no pair records, near-first stays)
        assert dev_s["probe_rays"] > 1000 and dev_s["visits_near"] > 1.0
    if dev_l["mode"] != 0 or scene == "DarkCornell":                            # (only scenes that live in LDS with few emitters probe the last rays)
        assert dev_l["mode"] == 1 + host_l["rule"] and dev_l["probe_rays"] == host_l["probe_rays"] > 500
        assert list(dev_l["probe_node_visits"].values()) == host_l["visits"]


def _with_emissive_materials(rpt, w, how_many_triangles):
    """a copy of the world in which the materials of the first few non-emissive triangles (in material order) emit as well: nee = 0 reads no light table"""
    import copy
    w2 = copy.copy(w)
    w2.materials = w.materials.copy()
    em = w2.materials["emissive"]
    tri_mat = w.indices["material"]
    have = int(sum((tri_mat == m).sum() for m in range(len(em)) if np.any(em[m, :3] != 0)))
    for m in np.argsort([int((tri_mat == m).sum()) for m in range(len(em))]):
        n = int((tri_mat == m).sum())
        if n == 0 or np.any(em[m, :3] != 0):
            continue
        if have + n > how_many_triangles:
            continue
        em[m] = (1.5, 0.25, 4.0, 0.0)
        have += n
    return w2, have


LAST_MODES = {"off": ("RPT_LAST_ORDER", "off", 0), "near": ("RPT_LAST_ORDER", "near", 1), "opaque": ("RPT_LAST_ORDER", "opaque", 2),
              "small": ("RPT_LAST_ORDER", "small", 3), "ratio": ("RPT_LAST_ORDER", "ratio", 4)}


@pytest.mark.parametrize("mode", sorted(LAST_MODES))
@pytest.mark.parametrize("case", ["cornell", "cornell_two_bounces", "cornell_roulette", "cornell_more_emitters", "cornell_too_many_emitters", "textured_open"])
def test_last_extension_rays_may_stop_at_their_first_hit(monkeypatch, hipmod, oracle, rpt, world, case, mode):
    """Without NEE the last extension ray of a path adds the sky on a miss, the emission of an emitter's front, and nothing otherwise (kernels/src/lib.rs:62-109):
    in a batch of known length the rays of that launch which pass the Moller-Trumbore test of no emissive triangle stop at their first accepted triangle
    (k_traverse_nearest_stream LAST), near child first or in one of three fixed orders over a flipped copy of the pair records — and with more than four
    emissive triangles, or switched off, they walk to the end.  Every mode gives the oracle's accumulators bit for bit: closed box, open textured scene
    with an image skybox, two bounces (the last ray is the second), roulette from the first bounce on, more emitters to test against."""
    var, val, expect = LAST_MODES[mode]
    monkeypatch.delenv("RPT_LAST_ORDER", raising=False)
    monkeypatch.setenv(var, val)
    skybox = None
    over = {}
    W, H, spp = 160, 96, 8
    if case == "textured_open":
        from scenes import textured_scene
        w, skybox = textured_scene()
        over = {"has_skybox": 1, "cam_position": (0.0, 1.6, -4.0, 0.0), "cam_rotation": (0.05, 0.1, 0.0, 0.0)}
    else:
        w = world("DarkCornell")
        if case == "cornell_two_bounces":
            over = {"min_bounces": 1, "max_bounces": 2}
        elif case == "cornell_roulette":
            over = {"min_bounces": 0, "max_bounces": 5}
        elif case == "cornell_more_emitters":
            w, n_em = _with_emissive_materials(rpt, w, 4)
            assert 2 < n_em <= 4
        elif case == "cornell_too_many_emitters":
            w, n_em = _with_emissive_materials(rpt, w, 40)
            assert n_em > 4
            expect = 0
    cfg = rpt.default_config(W, H, nee=0, **over)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w, skybox_f32=skybox)
        lo = r.last_bounce_order()
        if lo["mode"] != 0 or expect == 0:
            assert lo["mode"] == expect, lo
        r.set_config(cfg); r.reset(seeds)
        r.render(spp)
        acc_g, _ = r.read_accum()
        st_g = r.stats()
    finally:
        r.close()
    acc_c, _, st_c = oracle.trace_cpu(cfg, oracle.scene(w, skybox_f32=skybox), seeds, spp)
    assert st_g["extension_rays"] == st_c.extension_rays and st_g["sky_evals"] == st_c.sky_evals
    assert np.array_equal(acc_g.view(np.uint32), acc_c.view(np.uint32))


@pytest.mark.parametrize("scene,nee,over", [("DarkCornell", 0, {}), ("DarkCornell", 1, {"cam_position": (-0.0, 1.0, -3.5, 0.0)}), ("textured", 2, {"cam_position": (0.0, 1.6, -4.0, 0.0)}),
                                            ("DarkCornell", 0, {"cam_position": (0.0, 1e6, -3e7, 0.0)})])
def test_camera_rays_walk_planes_with_the_origin_already_subtracted(hipmod, oracle, rpt, world, scene, nee, over):
    """Iteration 0 of a render call traces camera rays only, all from cfg.cam_position: the streamed LDS walk of that launch stages `plane - origin`
    once per workgroup instead of computing it per lane and node (k_traverse_nearest_stream FIRST).  The image is the
    oracle's — with NEE, with a zero of either sign and with a far-away value in the camera position (the exact-division guard's other path), and for a
    call whose slots take more than one sample (only its first iteration is such a launch)."""
    skybox = None
    if scene == "textured":
        from scenes import textured_scene
        w, skybox = textured_scene()
    else:
        w = world(scene)
    W, H, spp = 136, 72, 40
    cfg = rpt.default_config(W, H, nee=nee, **over)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w, skybox_f32=skybox)
        r.set_config(cfg); r.reset(seeds)
        r.render(8)                      # a batch of known length
        r.render(spp - 8)                # slots take a second sample: regenerated paths share later launches with bounced ones
        acc_g, _ = r.read_accum()
        st_g = r.stats()
    finally:
        r.close()
    acc_c, _, st_c = oracle.trace_cpu(cfg, oracle.scene(w, skybox_f32=skybox), seeds, spp)
    assert st_g["extension_rays"] == st_c.extension_rays and st_g["shadow_rays"] == st_c.shadow_rays
    assert np.array_equal(acc_g.view(np.uint32), acc_c.view(np.uint32))


def test_the_library_chooses_the_order_of_the_last_rays_per_scene(monkeypatch, hipmod, rpt, world):
    """rpt_last_bounce_order after rpt_upload_scene: DarkCornell (two emissive triangles, lives in LDS) walks those rays in a fixed order its probe favoured;
    a scene that does not live in LDS keeps the whole walk; the decision is the host probe's (rpt_debug_last_order_host) and is re-taken on every upload."""
    monkeypatch.delenv("RPT_LAST_ORDER", raising=False)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(world("DarkCornell"))
        lo = r.last_bounce_order()
        host = hipmod.last_order_host(world("DarkCornell"))
        assert lo["mode"] == 1 + host["rule"] >= 2 and lo["emissive_triangles"] == 2 and lo["probe_rays"] == host["probe_rays"] > 500
        v = lo["probe_node_visits"]
        assert list(v.values()) == host["visits"] and min(list(v.values())[1:]) < 0.95 * v["near child first"]
        r.upload_scene(world("PBRTest"))
        assert r.last_bounce_order()["mode"] == 0
        r.upload_scene(world("DarkCornell"))
        assert r.last_bounce_order()["mode"] == lo["mode"]
    finally:
        r.close()


@pytest.mark.parametrize("nee,has_skybox", [(0, 1), (1, 1), (2, 0), (1, 0)])
def test_textured_scene_and_image_skybox_parity(renderer, oracle, rpt, nee, has_skybox):
    """Atlas sampling (CPU-polyfill semantics, image_polyfill.rs:32-55), normal mapping (lib.rs:132-141), uv wrap
    (lib.rs:127-129) and the image-skybox branch (lib.rs:70-78) — no shipped scene exercises these."""
    from scenes import textured_scene
    w, skybox = textured_scene()
    W, H, spp = 160, 96, 6
    cfg = rpt.default_config(W, H, nee=nee, has_skybox=has_skybox, cam_position=(0.0, 1.6, -4.0, 0.0),
                             cam_rotation=(0.05, 0.1, 0.0, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w, skybox_f32=skybox)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc_g, _ = renderer.read_accum()
    st_g = renderer.stats()
    acc_c, _, st_c = oracle.trace_cpu(cfg, oracle.scene(w, skybox_f32=skybox), seeds, spp)
    assert st_c.error_flags == 0 and st_c.sky_evals > 0 and st_g["sky_evals"] == st_c.sky_evals
    assert st_g["extension_rays"] == st_c.extension_rays and st_g["shadow_rays"] == st_c.shadow_rays
    err = rel_l2(acc_g[..., :3], acc_c[..., :3])
    assert err <= TOL_REL_L2
    assert np.array_equal(acc_g.view(np.uint32), acc_c.view(np.uint32))
    assert acc_c[..., :3].std() > 0.05          # the textures really modulate the image


def test_full_size_baseline_config_properties(hipmod, oracle, rpt, world):
    """BASELINE config[1] at full size (DarkCornell 1024x1024) through size-independent properties: every pixel got
    exactly spp samples, two runs agree bitwise, batch splitting is invisible, ray accounting is consistent, and a
    64x64 window of the full-size image equals the oracle's render of that window bit for bit."""
    W = H = 1024
    spp = 8
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    r.upload_scene(world("DarkCornell"))
    r.set_config(cfg)
    r.reset(seeds)
    r.render(spp)
    a, s = r.read_accum()
    st = r.stats()
    assert s == spp and np.all(a[..., 3] == spp) and np.isfinite(a).all()
    assert st["samples"] == W * H * spp and W * H * spp <= st["extension_rays"] <= W * H * spp * cfg.max_bounces
    r.reset(seeds)
    for n in (3, 5):
        r.render(n)
    b, _ = r.read_accum()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # windows in the middle, at two corners and across a 64-pixel tile seam: 16 M slots are in flight here, so this is
    # the streamed traversal with lane refill (k_traverse_nearest_stream) checked against the oracle at full size
    for rect in ((480, 470, 544, 534), (0, 0, 48, 40), (976, 992, 1024, 1024), (40, 600, 90, 650)):
        ref, _, _ = oracle.trace_cpu(cfg, oracle.scene(world("DarkCornell")), seeds, spp, rect=rect)
        x0, y0, x1, y1 = rect
        assert np.array_equal(a[y0:y1, x0:x1].view(np.uint32), ref[y0:y1, x0:x1].view(np.uint32)), rect
    r.close()


@pytest.mark.parametrize("scene,W,H,nee,spp", [("VeachMIS", 1920, 1080, 1, 3), ("PBRTest", 2048, 2048, 0, 2)])
def test_full_size_other_baseline_configs_windows(hipmod, oracle, rpt, world, scene, W, H, nee, spp):
    """BASELINE configs [2] and [3] at their full resolutions (few samples): windows of the full-size image equal the
    oracle's render of those windows bit for bit, and the ray counts obey their bounds."""
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    r.upload_scene(world(scene))
    r.set_config(cfg)
    r.reset(seeds)
    r.render(spp)
    a, s = r.read_accum()
    st = r.stats()
    assert s == spp and np.all(a[..., 3] == spp) and np.isfinite(a).all()
    assert W * H * spp <= st["extension_rays"] <= W * H * spp * cfg.max_bounces
    for rect in ((W // 2 - 24, H // 2 - 20, W // 2 + 24, H // 2 + 20), (0, H - 32, 40, H), (W - 40, 0, W, 36)):
        ref, _, _ = oracle.trace_cpu(cfg, oracle.scene(world(scene)), seeds, spp, rect=rect)
        x0, y0, x1, y1 = rect
        assert np.array_equal(a[y0:y1, x0:x1].view(np.uint32), ref[y0:y1, x0:x1].view(np.uint32)), rect
    r.close()


@pytest.mark.parametrize("scene,W,H,nee,spp", [("DarkCornell", 1024, 1024, 0, 256), ("VeachMIS", 1920, 1080, 1, 1024), ("PBRTest", 2048, 2048, 0, 512)])
def test_baseline_configs_at_their_own_sample_counts(hipmod, oracle, rpt, world, scene, W, H, nee, spp):
    """BASELINE configs [1], [2], [3] at their full resolution AND their full sample counts (256 / 1024 / 512 spp), rendered
    the way the reference's loop does — batches of sync_rate = 32 samples (src/trace.rs:75, 182-194): three windows of
    the final accumulators (centre, across a 64 x 64 tile corner, a far corner) equal the oracle's render of those windows
    at the same spp BIT FOR BIT — the f32 sum over all samples in sample order is part of the result
    (kernels/src/lib.rs:225-226) — every pixel carries exactly spp samples, rng[i].x == spp.

    Found by this test: at 1024 spp VeachMIS has a handful of NaN pixels IN THE REFERENCE'S SEMANTICS — the sky term is the one
    radiance contribution that is not wrapped in mask_nan (`radiance += throughput * skybox::scatter(..)`, lib.rs:69), and a
    glossy bounce with pdf = 0 leaves a NaN throughput (spectrum / pdf, lib.rs:168) that then escapes to the sky.  The oracle
    has the NaN in the same pixels from the same sample on; parity means reproducing it, so every non-finite pixel of the GPU
    image is checked against the oracle's value for that pixel (NaN for NaN), and they must stay rare."""
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    r.upload_scene(world(scene))
    r.set_config(cfg)
    r.reset(seeds)
    for _ in range(spp // 32):
        r.render_async(32)
    r.wait()
    a, s = r.read_accum()
    st = r.stats()
    rng = r.read_rng()
    assert s == spp and np.all(a[..., 3] == spp)
    assert np.all(rng["n"] == spp)
    assert st["samples"] == W * H * spp and W * H * spp <= st["extension_rays"] <= W * H * spp * cfg.max_bounces
    sc = oracle.scene(world(scene))
    bad = np.argwhere(~np.isfinite(a).all(axis=2))
    assert len(bad) <= 16, len(bad)                     # (measured: DarkCornell 0, VeachMIS 4 of 2 M pixels, PBRTest 0)
    for (y, x) in bad:
        ref, _, _ = oracle.trace_cpu(cfg, sc, seeds, spp, rect=(int(x), int(y), int(x) + 1, int(y) + 1))
        assert np.array_equal(a[y, x], ref[y, x], equal_nan=True), (int(x), int(y), a[y, x], ref[y, x])
    if scene == "DarkCornell":
        assert len(bad) == 0                            # a closed scene never reaches the unmasked sky term
    ww, wh = 48, 40
    ext = 0
    for (x0, y0) in ((W // 2 - ww // 2, H // 2 - wh // 2), (64 * (W // 192) - ww // 2, 64 * (H // 320) - wh // 2), (W - ww, H - wh)):
        rect = (x0, y0, x0 + ww, y0 + wh)
        ref, rng_ref, ost = oracle.trace_cpu(cfg, sc, seeds, spp, rect=rect)
        assert np.array_equal(a[y0:y0 + wh, x0:x0 + ww].view(np.uint32), ref[y0:y0 + wh, x0:x0 + ww].view(np.uint32)), rect
        ext += ost.extension_rays
    assert ext > ww * wh * spp                         # (the windows really traced: more than one ray per sample)
    r.close()


@pytest.mark.parametrize("scene,nee,spp", [("DarkCornell", 0, 7), ("VeachMIS", 1, 5), ("PBRTest", 2, 3)])
def test_samples_in_flight_invisible(hipmod, oracle, rpt, world, scene, nee, spp):
    """Any number of samples of a pixel in flight gives the sequential sample-order sum, bit for bit
    (k_complete.h) — including sample counts that are not a multiple of it."""
    w = world(scene)
    W, H = 96, 80
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    ref, rng_ref, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    for s_in_flight in (1, 2, 4, 16, 32, 256, 0):
        r = hipmod.Renderer(0)
        r.set_samples_in_flight(s_in_flight)
        r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
        r.render(spp - 2)
        r.render(2)
        acc, n = r.read_accum()
        assert n == spp
        assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"samples in flight = {s_in_flight}"
        g = r.stats()
        assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays
        assert np.array_equal(r.read_rng()["n"], rng_ref["n"])
        r.close()


@pytest.mark.parametrize("q_shift", [None, 5, 3])
@pytest.mark.parametrize("scene,nee", [("DarkCornell", 0), ("VeachMIS", 1)])
def test_up_to_256_samples_of_a_pixel_in_flight(monkeypatch, hipmod, oracle, rpt, world, scene, nee, q_shift):
    """More than 32 slots per pixel (round 6: k_complete counts a pixel's finished slots — a prefix — instead of keeping a 32-bit mask): 64, 128 and
    256 samples in flight give the sequential sample-order sum bit for bit (kernels/src/lib.rs:225-226), as batches of known length (n <= S),
    as calls whose slots take several samples (n > S), with sample counts that are no multiple of anything, under both slot layouts, and a
    context may move between them (the path state grows on demand)."""
    if q_shift is not None:
        monkeypatch.setenv("RPT_SLOT_Q_SHIFT", str(q_shift))
    w = world(scene)
    W, H = 56, 44                                                   # (2 464 pixels: not a multiple of 64)
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    calls = (256, 37, 130, 64, 200)                                 # 256 / 64 / 256 / 64 / 256 slots per pixel under S = 256
    ref, rng_ref, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, sum(calls))
    for s_in_flight in (256, 128, 64):
        r = hipmod.Renderer(0)
        try:
            r.set_samples_in_flight(s_in_flight)
            r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
            for k, n in enumerate(calls):
                (r.render_async if k % 2 == 0 else r.render)(n)
            r.wait()
            acc, n = r.read_accum()
            assert n == sum(calls)
            assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"samples in flight = {s_in_flight}"
            g = r.stats()
            assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays
            assert np.array_equal(r.read_rng()["n"], rng_ref["n"])
        finally:
            r.close()
    with pytest.raises(hipmod.RptError):
        r = hipmod.Renderer(0)
        try:
            r.set_samples_in_flight(257)
        finally:
            r.close()


def test_sky_stage_once_per_batch_or_per_iteration(hipmod, oracle, rpt, world):
    """A batch of known length shades its misses in ONE sky launch after its last iteration (a miss only ends a path, lib.rs:79); a call whose slots
    take several samples shades them in the iteration that found them (the slot must be free for its next sample).  Same image, same ray and sky
    counts either way, on the scene where most paths end in the sky."""
    w = world("PBRTest")
    W, H, spp = 128, 96, 8
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)
    ref, rng_ref, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    for in_flight in (0, 2):
        r = hipmod.Renderer(0)
        r.set_samples_in_flight(in_flight)
        r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
        r.render_async(5); r.render_async(3); r.wait()
        acc, n = r.read_accum()
        g = r.stats()
        assert n == spp and np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), in_flight
        assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays and g["sky_evals"] == st.sky_evals
        assert (g["kernel_launches"]["sky"] == 2) if in_flight == 0 else (g["kernel_launches"]["sky"] > 2 * cfg.max_bounces)
        r.close()


@pytest.mark.parametrize("q_shift", [1, 2, 3, 5])
def test_slot_layout_invisible(monkeypatch, hipmod, oracle, rpt, world, q_shift):
    """The slot layout (k_common.h slot_pix: a wave = 64 / Q pixels x Q samples; Q = 1 for the shipped scenes, 32 for scenes of
    half a million triangles and more) never reaches the image: every Q, sample counts that are not a multiple of the slots per
    pixel, a second call that continues the first, an image whose pixel count is not a multiple of 64."""
    monkeypatch.setenv("RPT_SLOT_Q_SHIFT", str(q_shift))
    w = world("VeachMIS")
    W, H, spp = 100, 70, 11
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)
    ref, rng_ref, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    for s_in_flight in (0, 8, 2):
        r = hipmod.Renderer(0)
        r.set_samples_in_flight(s_in_flight)
        r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
        r.render(spp - 3)
        r.render(3)
        acc, n = r.read_accum()
        assert n == spp
        assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), f"q_shift {q_shift}, samples in flight {s_in_flight}"
        g = r.stats()
        assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays
        assert np.array_equal(r.read_rng()["n"], rng_ref["n"])
        r.close()


def test_resolve_and_tonemap_parity(renderer, oracle, rpt, world):
    """mean + the six display tonemappers (render.wgsl:36-153) on the device == oracle, bitwise."""
    w = world("VeachMIS")
    W, H, spp = 160, 96, 6
    cfg = rpt.default_config(W, H, nee=1)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(rpt.blue_noise_seeds(W, H))
    renderer.render(spp)
    acc, n = renderer.read_accum()
    for op in range(7):
        got = renderer.resolve(op)
        want = oracle.resolve(acc, float(n), op)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"tonemap op {op}"
        if op in (1, 2, 3, 4):
            assert got.min() >= 0.0 and got.max() <= 1.0
    assert np.array_equal(renderer.resolve(0), acc[..., :3] / np.float32(n))


EDGE_CASES = [
    # W, H, spp, config overrides — degenerate sizes and bounce settings
    (1, 1, 5, {}),
    (3, 5, 4, {"nee": 1}),
    (65, 64, 3, {}),                                          # one pixel column spills into a second tile
    (40, 40, 3, {"max_bounces": 0}),                          # the bounce loop never runs: black image, w = spp
    (40, 40, 3, {"max_bounces": 1}),
    (40, 40, 4, {"min_bounces": 0, "max_bounces": 4}),        # roulette from bounce 1 on
    (40, 40, 4, {"min_bounces": 9, "max_bounces": 2, "nee": 2}),
    (40, 40, 3, {"nee": 7}),                                  # NextEventEstimation::from_u32(7) == None
    (40, 40, 3, {"specular_weight_clamp": (0.0, 1.0)}),
    (48, 32, 3, {"cam_position": (0.0, 1.0, 0.5, 0.0)}),     # camera inside the geometry
]


@pytest.mark.parametrize("W,H,spp,over", EDGE_CASES)
def test_edge_configurations(renderer, oracle, rpt, world, W, H, spp, over):
    for scene in ("DarkCornell", "VeachMIS"):
        w = world(scene)
        cfg = rpt.default_config(W, H, **over)
        seeds = rpt.blue_noise_seeds(W, H)
        renderer.upload_scene(w)
        renderer.set_config(cfg)
        renderer.reset(seeds)
        renderer.render(0)                                    # a zero-sample batch is a no-op
        renderer.render(spp)
        acc, n = renderer.read_accum()
        ref, rng_ref, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
        assert n == spp and st.error_flags == 0
        assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32)), (scene, over)
        assert np.array_equal(renderer.read_rng()["n"], rng_ref["n"])
        g = renderer.stats()
        assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays


def test_uniform_seed_mode(renderer, oracle, rpt, world):
    """rng_data_uniform (reference: src/trace.rs:158): n = random u32, offset = 0 — caller-supplied here."""
    W, H, spp = 64, 48, 4
    w = world("DarkCornell")
    cfg = rpt.default_config(W, H, nee=1)
    seeds = np.zeros(W * H, rpt._ffi.RNG_DTYPE)
    seeds["n"] = np.random.default_rng(99).integers(0, 2 ** 32, W * H, dtype=np.uint64).astype(np.uint32)
    seeds["n"][:4] = [0xFFFFFFFF, 0xFFFFFFFE, 0, 1]           # n + 1 wraps
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc, _ = renderer.read_accum()
    ref, rng_ref, _ = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(renderer.read_rng()["n"], rng_ref["n"])


@pytest.mark.parametrize("nee", [0, 1])
def test_deep_bvh_stand_in_parity(renderer, oracle, rpt, nee):
    """BASELINE config 5 names BreakTime.glb, which the reference mount lacks; the labelled stand-in is a procedural
    scene of clustered long thin triangles (tests/scenes.py): BVH depth ~20 (24-entry LDS stack variant, global-memory
    traversal), leaves of up to 64 triangles, ~80 node visits per ray."""
    from scenes import deep_bvh_scene
    w = deep_bvh_scene(50_000)
    assert w.bvh_max_depth >= 16 and w.nodes["triangle_count"].max() >= 32
    W, H, spp = 128, 96, 3
    cfg = rpt.default_config(W, H, nee=nee, cam_position=(0.0, 2.5, -0.5, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc, _ = renderer.read_accum()
    ref, _, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    g = renderer.stats()
    assert st.error_flags == 0 and st.max_stack >= 8
    assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))


def test_render_in_batches_equals_one_batch(renderer, rpt, world):
    w = world("DarkCornell")
    cfg = rpt.default_config(96, 96, nee=1)
    seeds = rpt.blue_noise_seeds(96, 96)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(6)
    a, _ = renderer.read_accum()
    renderer.reset(seeds)
    for n in (1, 2, 3):
        renderer.render(n)
    b, s = renderer.read_accum()
    assert s == 6 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_render_async_equals_render(rpt, hipmod, world):
    """rpt_render_async enqueues batches back to back without a host wait when the iteration count is known
    (n_samples <= slots per pixel) and falls back to the synchronous loop otherwise; accumulators, rng and ray counts
    must not depend on which entry point was used."""
    W, H = 96, 64
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)
    out = {}
    for mode in ("sync", "async"):
        r = hipmod.Renderer(0)
        r.upload_scene(world("DarkCornell"))
        r.set_config(cfg)
        r.reset(seeds)
        for n in (4, 16, 3, 40, 1):                  # 40 > slots per pixel: exercises the fall-back inside the async call
            (r.render if mode == "sync" else r.render_async)(n)
        if mode == "async":
            r.wait()
        acc, ns = r.read_accum()
        st = r.stats()
        out[mode] = (acc.copy(), ns, r.read_rng().copy(), st["extension_rays"], st["shadow_rays"])
        r.close()
    assert out["sync"][1] == out["async"][1] == 64
    assert np.array_equal(out["sync"][0].view(np.uint32), out["async"][0].view(np.uint32))
    assert np.array_equal(out["sync"][2], out["async"][2])
    assert out["sync"][3:5] == out["async"][3:5]


def test_resume_from_mean_times_samples(renderer, oracle, rpt, world):
    """accum_init = mean * samples (reference: src/trace.rs:163-164)."""
    w = world("FurnaceTest")
    cfg = rpt.default_config(64, 64)
    seeds = rpt.blue_noise_seeds(64, 64)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    init = np.full((64, 64, 4), 0.25, np.float32) * np.float32(4.0)
    init[..., 3] = 4.0
    renderer.reset(seeds, accum_init=init, samples_init=4)
    renderer.render(2)
    acc, samples = renderer.read_accum()
    acc_c, _, _ = oracle.trace_cpu(cfg, oracle.scene(w), seeds, 2, accum=init)
    assert samples == 6
    assert np.array_equal(acc.view(np.uint32), acc_c.view(np.uint32))


def test_tile_partition_is_invisible(hipmod, rpt, world, tiles):
    """Rendering as rank r of 3 gives exactly the pixels of the full render (8e: determinism across GPU counts)."""
    w = world("DarkCornell")
    W, H = 200, 136
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)
    full = hipmod.Renderer(0)
    full.upload_scene(w); full.set_config(cfg); full.reset(seeds); full.render(3)
    ref, _ = full.read_accum()
    full.close()
    blocks = []
    for r in range(3):
        part = hipmod.Renderer(0, rank=r, world_size=3)
        part.upload_scene(w); part.set_config(cfg); part.reset(seeds); part.render(3)
        img, _ = part.read_accum()
        blk = tiles.tile_block_from_image(img, r, 3)
        assert part.local_pixels() == len(blk) == part.rank_pixels(r)
        blocks.append(blk)
        own = np.zeros((H, W), bool)
        xy = hipmod.tile_order(W, H, r, 3)
        own[xy >> 16, xy & 0xFFFF] = True
        assert np.all(img[~own] == 0)
        part.close()
    assert np.array_equal(tiles.untile_host(blocks, W, H, 3).view(np.uint32), ref.view(np.uint32))


def test_device_untile_matches_host(hipmod, rpt, world, tiles):
    import torch
    w = world("DarkCornell")
    W, H = 136, 72
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    parts, blocks = [], []
    for r in range(2):
        part = hipmod.Renderer(0, rank=r, world_size=2)
        part.upload_scene(w); part.set_config(cfg); part.reset(seeds); part.render(2)
        blocks.append(tiles.device_block_as_tensor(part, "cuda:0").clone())
        parts.append(part)
    cat = torch.cat(blocks, 0).contiguous()
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    parts[0].untile(cat.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    host = tiles.untile_host([b.cpu().numpy() for b in blocks], W, H, 2)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), host.view(np.uint32))
    # strided form (equal-sized gather slots, as bench.py uses)
    stride = max(len(b) for b in blocks) + 5
    padded = torch.full((2, stride, 4), float("nan"), dtype=torch.float32, device="cuda:0")
    for r_, b in enumerate(blocks):
        padded[r_, : len(b)] = b
    out2 = torch.zeros_like(out)
    parts[0].untile(padded.data_ptr(), out2.data_ptr(), stride)
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy().view(np.uint32), host.view(np.uint32))
    for p in parts:
        p.close()


def test_host_dispatch_trace_gpu_furnace(rpt):
    """The reference's own test, through the host-side mirror: furnace_test(use_cpu=false, use_mis) —
    tests/correctness_tests.rs:14-33 — at the reference's settings (128^2, 32 spp, pixel (65,75))."""
    for use_mis in (False, True):
        state = rpt.setup_trace(128, 128, 32)
        if use_mis:
            state.config.nee = 1
        rpt.trace_gpu(rpt.fixture("FurnaceTest.glb"), None, state)
        assert state.samples == 32
        frame = state.framebuffer()
        px = frame[75, 65] ** (1.0 / 2.2)
        assert np.all(np.abs(px - 0.8) < 0.02), px
        state.close()


@pytest.mark.parametrize("knob", ["RPT_NO_LDS_SCENE=1", "RPT_SHADE_COMPACT=1", "RPT_SHADE_COMPACT=0", "RPT_STAGE_TIMING=1", "RPT_STAGE_TIMING=2", "RPT_UPLOAD_TIMING=1",
                                  "RPT_SKY_STRIDED=3", "RPT_SKY_STRIDED=1", "RPT_SKY_STRIDED=0", "RPT_STACK_BITS=21", "RPT_STACK_BITS=24", "RPT_STACK_BITS=32",
                                  "RPT_COOP_LEAVES=1", "RPT_SLOT_Q_SHIFT=3", "RPT_SHADOW_ORDER=near", "RPT_SHADOW_ORDER=fixed", "RPT_LAST_ORDER=off"])
def test_developer_knobs_do_not_change_the_image(monkeypatch, hipmod, rpt, world, knob):
    """Every environment variable the library reads (rpt_ctx.h rpt_knobs — one place, eleven of them + the test stand-in for RCCL) leaves the image
    bit-identical: they select kernels / schedules / diagnostics, never arithmetic."""
    W, H, spp = 160, 96, 6
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)

    # (the global-memory walk / a scene open to the sky / the LDS walk)
    scene = "VeachMIS" if knob.startswith(("RPT_STACK_BITS", "RPT_COOP_LEAVES")) else "PBRTest" if knob.startswith("RPT_SKY_STRIDED") else "DarkCornell"

    def render():
        r = hipmod.Renderer(0)
        r.upload_scene(world(scene))
        r.set_config(cfg)
        r.reset(seeds)
        r.render(spp)
        acc, _ = r.read_accum()
        st = r.stats()
        r.close()
        return acc, (st["extension_rays"], st["shadow_rays"], st["sky_evals"])

    base = render()
    for one in knob.split(","):
        name, value = one.split("=")
        monkeypatch.setenv(name, value)
    got = render()
    assert got[1] == base[1]
    assert np.array_equal(got[0].view(np.uint32), base[0].view(np.uint32))


@pytest.mark.parametrize("scene,nee", [("PBRTest", 0), ("VeachMIS", 1), ("DarkCornell", 2)])
def test_packed_shade_stage_equals_the_oracle(monkeypatch, hipmod, oracle, rpt, world, scene, nee):
    """k_shade<.., COMPACT> (traversed slots packed per workgroup, generations completed at the start of the next pass)
    forced on: several render calls, one of them with more samples than slots per pixel (so finished generations restart
    from the completion step), ragged image — accumulators, rng and ray counts equal the oracle's."""
    monkeypatch.setenv("RPT_SHADE_COMPACT", "1")
    W, H = 200, 136
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    r.set_samples_in_flight(4)
    r.upload_scene(world(scene))
    r.set_config(cfg)
    r.reset(seeds)
    for n in (3, 9, 1):
        r.render(n)
    r.render_async(4)
    r.wait()
    acc, ns = r.read_accum()
    st = r.stats()
    ref, rng_ref, so = oracle.trace_cpu(cfg, oracle.scene(world(scene)), seeds, 17)
    assert ns == 17 and st["extension_rays"] == so.extension_rays and st["shadow_rays"] == so.shadow_rays
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(r.read_rng()["n"], np.full(W * H, 17, np.uint32))
    r.close()


def test_error_behaviour(hipmod, rpt, world):
    r = hipmod.Renderer(0)
    with pytest.raises(hipmod.RptError):           # render before scene/config
        r.render(1)
    r.upload_scene(world("DarkCornell"))
    bad = rpt.default_config(64, 64, nee=1, max_bounces=5)     # 2 + 5*7 > 31 LDS dimensions: reference panics
    with pytest.raises(hipmod.RptError) as e:
        r.set_config(bad)
    assert e.value.code == -4
    r.close()


@pytest.mark.parametrize("seed", [1234, 77])
def test_randomised_configurations_equal_the_oracle(seed):
    """A fixed-seed slice of tools/fuzz_parity.py: random scene, image size (ragged, down to 1 pixel), spp split over two
    render calls, NEE mode, bounce limits, samples in flight, camera (some far outside the atmosphere or under the ground),
    sun direction / intensity, lobe-pick clamp — accumulators and the three ray counters equal the oracle's, bit for bit."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.main(n_cases=14, seed=seed) == 0


@pytest.mark.parametrize("nee,n_stack", [(0, 220), (1, 220), (1, 300)])
def test_unsplittable_fat_leaf_parity(renderer, oracle, rpt, nee, n_stack):
    """n_stack triangles with one centroid stay ONE leaf (tests/scenes.py fat_leaf_scene): too fat for the LDS image.  220: the streamed
    global-memory walk with the wave-cooperative leaf test over several rounds of 64 lanes; 300: more than the 254 triangles a link of the
    walk's pair records can count (k_traverse.h SceneViewPairsT), so the scene keeps the one-shot generic walks over the reference's own node
    array — image, rng and ray counts equal the oracle's, and so does every single ray."""
    from scenes import fat_leaf_scene
    w = fat_leaf_scene(n_stack)
    assert w.nodes["triangle_count"].max() == n_stack
    W, H, spp = 96, 80, 4
    cfg = rpt.default_config(W, H, nee=nee, cam_position=(0.0, 1.4, -0.8, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc, _ = renderer.read_accum()
    ref, rng_ref, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    g = renderer.stats()
    assert st.error_flags == 0
    assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(renderer.read_rng()["n"], rng_ref["n"])
    # and ray by ray: nearest and any-hit from random origins
    rng = np.random.default_rng(5)
    o, d = _random_rays(rng, 20000, w)
    sc = oracle.scene(w)
    t_g, tri_g, fl_g = renderer.debug_trace_rays(False, o, d)
    t_c, tri_c, fl_c, err = oracle.trace_rays(sc, 0, o, d)
    hit = (fl_c & 1) == 1
    assert err == 0 and np.array_equal(fl_g, fl_c) and hit.sum() > 1000
    assert np.array_equal(t_g.view(np.uint32), t_c.view(np.uint32)) and np.array_equal(tri_g[hit], tri_c[hit])
    max_t = (rng.random(len(o)) * 6).astype(np.float32)
    _, _, afl_g = renderer.debug_trace_rays(True, o, d, max_t)
    _, _, afl_c, err = oracle.trace_rays(sc, 1, o, d, max_t)
    assert err == 0 and np.array_equal(afl_g & 1, afl_c & 1)


@pytest.mark.parametrize("seed", [3, 4])
def test_random_api_sequences_stay_consistent_with_the_oracle(seed):
    """tools/fuzz_api.py: one context driven through a random valid sequence of ABI calls (render / render_async of 0..7
    samples, reads, new configurations, resets incl. resume, other scenes, samples-in-flight changes, the local
    communicator's gather) while a model advances the CPU oracle by the same samples: every read equals the model."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_api", os.path.join(root, "tools", "fuzz_api.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.main(steps=160, seed=seed, quiet=True) == 0


@pytest.mark.parametrize("nee", [0, 1])
def test_more_than_65536_nodes_parity(renderer, oracle, rpt, nee):
    """60 k small scattered triangles -> 120 003 nodes (tests/scenes.py scatter_scene): the 32-bit stack entries of the
    global-memory walks, which no shipped scene reaches — image, rng, ray counts and single rays equal the oracle's."""
    from scenes import scatter_scene
    w = scatter_scene()
    assert len(w.nodes) > 65536
    W, H, spp = 112, 80, 3
    cfg = rpt.default_config(W, H, nee=nee, cam_position=(0.0, 1.8, -0.9, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc, _ = renderer.read_accum()
    sc = oracle.scene(w)
    ref, rng_ref, st = oracle.trace_cpu(cfg, sc, seeds, spp)
    g = renderer.stats()
    assert st.error_flags == 0 and st.max_stack >= 8
    assert g["extension_rays"] == st.extension_rays and g["shadow_rays"] == st.shadow_rays
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(renderer.read_rng()["n"], rng_ref["n"])
    rng = np.random.default_rng(6)
    o, d = _random_rays(rng, 50000, w)
    t_g, tri_g, fl_g = renderer.debug_trace_rays(False, o, d)
    t_c, tri_c, fl_c, err = oracle.trace_rays(sc, 0, o, d)
    hit = (fl_c & 1) == 1
    assert err == 0 and np.array_equal(fl_g, fl_c) and hit.sum() > 1000
    assert np.array_equal(t_g.view(np.uint32), t_c.view(np.uint32)) and np.array_equal(tri_g[hit], tri_c[hit])
    max_t = (rng.random(len(o)) * 6).astype(np.float32)
    _, _, afl_g = renderer.debug_trace_rays(True, o, d, max_t)
    _, _, afl_c, err = oracle.trace_rays(sc, 1, o, d, max_t)
    assert err == 0 and np.array_equal(afl_g & 1, afl_c & 1)


@pytest.mark.parametrize("knobs", ["RPT_SHADE_COMPACT=1", "RPT_SKY_STRIDED=3", "RPT_SKY_STRIDED=2", "RPT_SHADE_COMPACT=1,samples_in_flight=2"])
@pytest.mark.parametrize("nee,has_skybox", [(0, 1), (1, 1), (2, 0)])
def test_kernel_variants_on_the_textured_scene(monkeypatch, hipmod, oracle, rpt, knobs, nee, has_skybox):
    """The template instantiations no shipped scene reaches together: TEXTURED x packed shade stage, and the strided sky
    stage on the image-skybox branch (lib.rs:70-78) as well as on the procedural sky — forced on, against the oracle."""
    from scenes import textured_scene
    in_flight = 0
    for one in knobs.split(","):
        name, value = one.split("=")
        if name == "samples_in_flight":
            in_flight = int(value)
        else:
            monkeypatch.setenv(name, value)
    w, skybox = textured_scene()
    W, H, spp = 136, 88, 5
    cfg = rpt.default_config(W, H, nee=nee, has_skybox=has_skybox, cam_position=(0.0, 1.6, -4.0, 0.0), cam_rotation=(0.05, 0.1, 0.0, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    r.set_samples_in_flight(in_flight)
    r.upload_scene(w, skybox_f32=skybox)
    r.set_config(cfg)
    r.reset(seeds)
    r.render(2)
    r.render(spp - 2)
    acc, n = r.read_accum()
    st = r.stats()
    r.close()
    ref, _, so = oracle.trace_cpu(cfg, oracle.scene(w, skybox_f32=skybox), seeds, spp)
    assert n == spp and so.error_flags == 0 and so.sky_evals > 0
    assert (st["extension_rays"], st["shadow_rays"], st["sky_evals"]) == (so.extension_rays, so.shadow_rays, so.sky_evals)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))


def test_deep_tree_with_wide_stack_entries(renderer, oracle, rpt):
    """300 k scattered triangles: 599 965 nodes, depth 24 — the 32-entry stack with entries wider than 16 bits (16 in LDS + 5
    mask registers, WaveStack<32, 21>; RPT_STACK_BITS forces the 24- and 32-bit forms elsewhere), the shape a real large scene
    has; image and ray counts against the oracle."""
    from scenes import scatter_scene
    w = scatter_scene(300_000)
    assert len(w.nodes) > 65536 and w.bvh_max_depth >= 24
    W, H, spp = 80, 56, 2
    cfg = rpt.default_config(W, H, nee=1, cam_position=(0.0, 1.8, -0.9, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w); renderer.set_config(cfg); renderer.reset(seeds)
    renderer.render(spp)
    acc, _ = renderer.read_accum()
    st = renderer.stats()
    ref, _, so = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    assert so.max_stack >= 10 and (st["extension_rays"], st["shadow_rays"]) == (so.extension_rays, so.shadow_rays)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))


def test_one_ray_per_lane_walk_of_a_foreign_node_pool(hipmod, oracle, rpt):
    """A tree whose children are not the nodes (2p + 1, 2p + 2) of a pair — no pool of the reference's builder, but a valid buffer of the boundary — keeps the
    generic walks (k_traverse_nearest / k_traverse_shadow, one ray per lane, 32-bit stack entries): the scene of more than 65 536 nodes, NEE on."""
    from scenes import scatter_scene, foreign_pool
    w = foreign_pool(scatter_scene())
    W, H, spp = 96, 64, 2
    cfg = rpt.default_config(W, H, nee=1, cam_position=(0.0, 1.8, -0.9, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
    r.render(spp)
    acc, _ = r.read_accum()
    st = r.stats()
    r.close()
    ref, _, so = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    assert (st["extension_rays"], st["shadow_rays"]) == (so.extension_rays, so.shadow_rays)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))


def test_contexts_on_different_threads(hipmod, oracle, rpt, world):
    """include/rpt/rpt.h: a context is single-caller, but DIFFERENT contexts may be driven from different threads (ctypes
    releases the GIL for the duration of a call): four threads, each with its own context, scene and configuration, render
    concurrently on the one GPU; every image equals the oracle's."""
    import threading
    jobs = [("DarkCornell", 120, 72, 0, 5), ("VeachMIS", 96, 80, 1, 4), ("DarkCornell", 64, 100, 2, 6), ("FurnaceTest", 80, 64, 1, 3)]
    out, errors = [None] * len(jobs), []

    def run(i):
        try:
            scene, W, H, nee, spp = jobs[i]
            cfg = rpt.default_config(W, H, nee=nee)
            seeds = rpt.blue_noise_seeds(W, H)
            r = hipmod.Renderer(0)
            r.upload_scene(world(scene)); r.set_config(cfg); r.reset(seeds)
            for _ in range(spp):
                r.render_async(1)
            r.wait()
            r.render(spp)
            out[i] = (r.read_accum()[0], r.stats())
            r.close()
        except Exception as e:                                   # noqa: BLE001
            errors.append((i, repr(e)))

    for name in {j[0] for j in jobs}:
        world(name)                                             # (load the scenes once, outside the threads)
    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (scene, W, H, nee, spp), (img, st) in zip(jobs, out):
        cfg = rpt.default_config(W, H, nee=nee)
        ref, _, so = oracle.trace_cpu(cfg, oracle.scene(world(scene)), rpt.blue_noise_seeds(W, H), 2 * spp)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), scene
        assert (st["extension_rays"], st["shadow_rays"]) == (so.extension_rays, so.shadow_rays)


def test_contexts_can_be_destroyed_with_work_in_flight_and_leak_nothing(hipmod, rpt, world):
    """rpt_destroy with an asynchronous batch (and a gather) still in flight completes it first; 60 create / upload /
    configure / render / destroy cycles give every byte of device memory back."""
    import torch
    w = world("DarkCornell")
    W, H = 256, 192
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)

    def cycle(k):
        r = hipmod.Renderer(0)
        if k % 3 == 0:
            r.comm_init_local()
        r.upload_scene(w); r.set_config(cfg); r.reset(seeds)
        r.render_async(8)
        if k % 3 == 0:
            r.gather_async()
        if k % 2 == 0:
            r.render_async(3)
        r.close()                                               # nothing waited for

    cycle(0)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(60):
        cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free1 - free0) <= (8 << 20), (free0, free1)


def test_misuse_is_an_error_code_never_a_crash(hipmod, rpt, world):
    """§8b: no aborts across the ABI — every out-of-order or out-of-range call returns a negative code with a message, and the
    context stays usable afterwards."""
    import ctypes as C
    L = hipmod.lib()
    w = world("DarkCornell")
    W, H = 72, 40
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    with pytest.raises(hipmod.RptError) as e:
        r.gather_async()                                        # no communicator
    assert e.value.code < 0 and "communicator" in str(e.value)
    assert L.rpt_read_gathered(r._h, None, None) != 0 and L.rpt_comm_world(r._h, None, None) != 0 and L.rpt_gather_wait(r._h) != 0
    assert L.rpt_set_partition(r._h, 3, 3) != 0 and L.rpt_set_partition(r._h, 0, 0) != 0
    assert L.rpt_render(None, 1) != 0 and L.rpt_wait(None) != 0 and L.rpt_set_config(r._h, None) != 0
    r.upload_scene(w)
    with pytest.raises(hipmod.RptError):
        r.render(1)                                             # no config yet
    r.set_config(cfg)
    assert L.rpt_reset(r._h, None, None, 0) != 0                # null seeds
    zero = rpt.default_config(0, 10)
    with pytest.raises(hipmod.RptError):
        r.set_config(zero)
    r.set_config(cfg)
    r.reset(seeds)
    with pytest.raises(hipmod.RptError):
        r.gather_async()                                        # still no communicator
    r.comm_init_local()
    assert L.rpt_read_gathered(r._h, np.zeros((H, W, 4), np.float32).ctypes.data_as(C.c_void_p), None) != 0   # nothing gathered yet
    bad, first = C.c_uint64(), C.c_uint32()
    assert L.rpt_debug_math_sweep(r._h, 7, 0, 1, C.c_float(1.0), C.byref(bad), C.byref(first)) != 0           # unknown op
    assert L.rpt_debug_math_sweep(r._h, 0, 0, (1 << 32) + 1, C.c_float(1.0), C.byref(bad), C.byref(first)) != 0
    # ... and the context still works
    r.render(3)
    img, n = r.read_accum()
    ref = hipmod.Renderer(0)
    ref.upload_scene(w); ref.set_config(cfg); ref.reset(seeds); ref.render(3)
    assert n == 3 and np.array_equal(img.view(np.uint32), ref.read_accum()[0].view(np.uint32))
    ref.close(); r.close()
    for devs, flags in (([], 0), ([99], 0), ([0, 0], 0)):
        with pytest.raises(hipmod.RptError):
            hipmod.MultiRenderer(devs, allow_shared_device=bool(flags))


def test_malformed_scene_buffers_are_rejected_not_traversed(hipmod, oracle, rpt, world):
    """rpt_upload_scene takes raw buffers from its host: an index, material, light-table or BVH entry that points outside its
    array, a BVH that is not a tree, a texture flag without an atlas — RPT_ESCENE with a message, no kernel ever sees them; the
    context then takes a valid scene and renders it correctly."""
    import copy
    base = world("DarkCornell")
    r = hipmod.Renderer(0)

    def broken(mutate):
        w = copy.copy(base)
        for name in ("per_vertex", "indices", "nodes", "materials", "light_pick"):
            setattr(w, name, getattr(base, name).copy())
        mutate(w)
        with pytest.raises(hipmod.RptError) as e:
            r.upload_scene(w)
        assert e.value.code == -5 and len(str(e.value)) > 20, str(e.value)

    nv, nt, nn = len(base.per_vertex), len(base.indices), len(base.nodes)
    inner = int(np.nonzero(base.nodes["triangle_count"] == 0)[0][1])
    leaf = int(np.nonzero(base.nodes["triangle_count"] > 0)[0][0])
    broken(lambda w: w.indices["v1"].__setitem__(5, nv))
    broken(lambda w: w.indices["material"].__setitem__(0, len(base.materials)))
    broken(lambda w: w.light_pick["triangle_index_a"].__setitem__(0, nt + 7))
    broken(lambda w: w.nodes["left_or_first"].__setitem__(inner, nn - 1))          # right child = nn: out of bounds
    broken(lambda w: w.nodes["left_or_first"].__setitem__(leaf, nt))               # leaf range past the triangles
    broken(lambda w: w.nodes["left_or_first"].__setitem__(inner, 0))               # a cycle through the root
    broken(lambda w: w.materials["has_albedo_texture"].__setitem__(0, 1))          # no atlas supplied
    W, H = 80, 56
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)
    r.upload_scene(base); r.set_config(cfg); r.reset(seeds); r.render(3)
    ref, _, _ = oracle.trace_cpu(cfg, oracle.scene(base), seeds, 3)
    assert np.array_equal(r.read_accum()[0].view(np.uint32), ref.view(np.uint32))
    r.close()


def test_nan_and_infinite_inputs_follow_the_reference_too():
    """tools/nan_probe.py: NaN / infinite / zero camera, rotation, sun and lobe-clamp values, a NaN vertex, a NaN box —
    nothing hangs, and image (NaN for NaN) and ray counts equal the oracle's on the LDS walk and the global-memory walk."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "nan_probe.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " ok" in l or "MISMATCH" in l]
    assert len(lines) == 38 and not any("MISMATCH" in l for l in lines), out.stdout[-2000:]


def test_one_triangle_scenes_and_scenes_without_lights():
    """tools/tiny_scene_probe.py: 1 / 2 / 3 / 5 triangles (a root that is a leaf; the LDS image of a single pair), no emissive
    triangle at all (the light table's sentinel entry) with every NEE mode, everything emissive — 36 cases against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "tiny_scene_probe.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


def test_hostile_texture_coordinates():
    """tools/uv_probe.py: NaN / infinite / huge / negative / exactly-one / denormal uvs on the textured scene, atlas and image
    skybox: addressing follows the oracle's casts (image_polyfill.rs:38-55) and stays inside the atlas."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "uv_probe.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


def test_hostile_material_values():
    """tools/material_probe.py: roughness / metallic / albedo / emissive set to 0, 1, tiny, huge, negative, NaN, infinite in
    random combinations: NaN throughput, infinite pdfs and all — the HIP path makes of them what the oracle does."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "material_probe.py"), "40", "5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]


def test_resolve_on_hostile_accumulators(renderer, oracle, rpt, world):
    """rpt_resolve (mean + the six tonemappers, render.wgsl:36-153) on accumulators holding NaN, infinities, negative, huge,
    denormal and zero values (loaded through the resume path): device == oracle, NaN for NaN; pow / division inside the
    curves included.  Tonemap op out of range: an error."""
    W, H = 64, 40
    cfg = rpt.default_config(W, H)
    renderer.upload_scene(world("DarkCornell"))
    renderer.set_config(cfg)
    rng = np.random.default_rng(8)
    vals = np.array([0.0, -0.0, 1.0, 0.18, 1e-42, 1e-30, 1e30, 3e38, -1.0, -1e30, np.nan, np.inf, -np.inf, 7.5, 0.999], np.float32)
    init = vals[rng.integers(0, len(vals), (H, W, 4))].copy()
    init[..., 3] = 5.0
    renderer.reset(rpt.blue_noise_seeds(W, H), accum_init=init, samples_init=5)
    acc, n = renderer.read_accum()
    assert n == 5
    for op in range(7):
        got = renderer.resolve(op)
        want = oracle.resolve(acc, float(n), op)
        ng, nw = np.isnan(got), np.isnan(want)
        assert np.array_equal(ng, nw) and np.array_equal(got[~ng].view(np.uint32), want[~nw].view(np.uint32)), f"tonemap op {op}"
    with pytest.raises(Exception):
        renderer.resolve(9)


def test_images_at_the_coordinate_limits():
    """tools/skinny_image_probe.py: 65535 x 1, 1 x 65535, 65535 x 2, 3 x 40000, 70 x 65535 as one rank and as three ranks whose
    partial images add up to the oracle's; 65536 and 0 are refused by rpt_set_config."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "skinny_image_probe.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout and "ACCEPTED" not in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]
