"""BASELINE config[3] AS IT IS WRITTEN: "PBRTest.glb with albedo/normal/rough/metal textures 2048x2048" (tile-split over 8 GPUs).

The shipped PBRTest.glb carries no texture, so the configuration is exercised on the file's own buffers plus a
labelled synthetic 4096 x 4096 RGBA8 atlas laid out by the reference's packer (tests/scenes.py pbrtest_textured_scene).  What must
match: the CPU polyfill sampler (shared_structs/src/image_polyfill.rs:32-55: texel = u8 / 255, alpha 1, wrap by `as usize %`,
floor / ceil footprint, three lerps), get_pbr_bsdf's three lookups (kernels/src/bsdf.rs:354-387) and the normal-map branch
(kernels/src/lib.rs:132-141) — at the configuration's own resolution, through the C ABI, bit for bit against the oracle.
"""
import numpy as np
import pytest

from scenes import pbrtest_textured_scene

pytestmark = pytest.mark.gpu

_cache = {}


def textured_world():
    if "w" not in _cache:
        _cache["w"] = pbrtest_textured_scene()
    return _cache["w"]


def _windows(W, H, ww=48, wh=40):
    # centre (spheres), across a 64 x 64 tile corner (two ranks' tiles when the image is split), a far corner, and one over the sphere grid's edge
    return ((W // 2 - ww // 2, H // 2 - wh // 2), (64 * (W // 192) - ww // 2, 64 * (H // 320) - wh // 2), (W - ww, H - wh), (W // 3, (2 * H) // 3))


@pytest.mark.parametrize("nee,spp", [(0, 4), (1, 3), (2, 2)])
def test_textured_pbrtest_at_2048_windows_equal_the_oracle(hipmod, oracle, rpt, world, nee, spp):
    W = H = 2048
    w = textured_world()
    cfg = rpt.default_config(W, H, nee=nee)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w)
        r.set_config(cfg)
        r.reset(seeds)
        r.render(spp)
        a, s = r.read_accum()
        st = r.stats()
        assert s == spp and np.all(a[..., 3] == spp)
        assert W * H * spp <= st["extension_rays"] <= W * H * spp * cfg.max_bounces
        sc = oracle.scene(w)
        ext = shadow = 0
        for (x0, y0) in _windows(W, H):
            rect = (x0, y0, x0 + 48, y0 + 40)
            ref, _, ost = oracle.trace_cpu(cfg, sc, seeds, spp, rect=rect)
            assert ost.error_flags == 0
            assert np.array_equal(a[y0:y0 + 40, x0:x0 + 48].view(np.uint32), ref[y0:y0 + 40, x0:x0 + 48].view(np.uint32)), rect
            ext += ost.extension_rays
            shadow += ost.shadow_rays
        assert ext > 4 * 48 * 40 * spp                     # the windows hit surfaces (more than one ray per sample)
        assert shadow == 0                                   # (PBRTest.glb has no emitter: the light table is the sentinel, NEE adds nothing)
        if nee == 0:
            # the textures really reach the image: the untextured file renders differently in the window over the spheres
            r.upload_scene(world("PBRTest"))
            r.reset(seeds)
            r.render(spp)
            b, _ = r.read_accum()
            x0, y0 = _windows(W, H)[0]
            assert not np.array_equal(a[y0:y0 + 40, x0:x0 + 48], b[y0:y0 + 40, x0:x0 + 48])
    finally:
        r.close()


def test_textured_pbrtest_at_its_own_sample_count(hipmod, oracle, rpt):
    """The configuration's 512 spp in the reference's batches of 32 (src/trace.rs:75): every pixel carries 512 samples, rng.n == 512, three
    windows of the final accumulators equal the oracle's bit for bit (the f32 sum over all samples in sample order is part of the result)."""
    W = H = 2048
    spp = 512
    w = textured_world()
    cfg = rpt.default_config(W, H)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w)
        r.set_config(cfg)
        r.reset(seeds)
        for _ in range(spp // 32):
            r.render_async(32)
        r.wait()
        a, s = r.read_accum()
        rng = r.read_rng()
        st = r.stats()
        assert s == spp and np.all(a[..., 3] == spp) and np.all(rng["n"] == spp)
        assert st["samples"] == W * H * spp
        sc = oracle.scene(w)
        bad = np.argwhere(~np.isfinite(a).all(axis=2))
        assert len(bad) <= 16, len(bad)
        for (y, x) in bad:
            ref, _, _ = oracle.trace_cpu(cfg, sc, seeds, spp, rect=(int(x), int(y), int(x) + 1, int(y) + 1))
            assert np.array_equal(a[y, x], ref[y, x], equal_nan=True), (int(x), int(y))
        for (x0, y0) in _windows(W, H)[:3]:
            ref, _, _ = oracle.trace_cpu(cfg, sc, seeds, spp, rect=(x0, y0, x0 + 48, y0 + 40))
            assert np.array_equal(a[y0:y0 + 40, x0:x0 + 48].view(np.uint32), ref[y0:y0 + 40, x0:x0 + 48].view(np.uint32)), (x0, y0)
    finally:
        r.close()


def test_textured_pbrtest_tile_split_is_invisible(hipmod, rpt):
    """The configuration is "tile-split 8 x MI355X": the 8 ranks' blocks, rendered one after the other on this GPU and un-tiled, are the
    one-rank image bit for bit (the atlas is replicated like the rest of the scene; pixels are independent, lib.rs:209-226)."""
    W, H, spp = 512, 384, 3
    w = textured_world()
    cfg = rpt.default_config(W, H, nee=1)
    seeds = rpt.blue_noise_seeds(W, H)
    r = hipmod.Renderer(0)
    try:
        r.upload_scene(w)
        r.set_config(cfg)
        r.reset(seeds)
        r.render(spp)
        whole, _ = r.read_accum()
    finally:
        r.close()
    out = np.zeros_like(whole)
    for rank in range(8):
        rr = hipmod.Renderer(0, rank=rank, world_size=8)
        try:
            rr.upload_scene(w)
            rr.set_config(cfg)
            rr.reset(seeds)
            rr.render(spp)
            part, _ = rr.read_accum()
            xy = hipmod.tile_order(W, H, rank, 8)
            xs, ys = (xy & 0xffff).astype(np.int64), (xy >> 16).astype(np.int64)
            out[ys, xs] = part[ys, xs]
        finally:
            rr.close()
    assert np.array_equal(out.view(np.uint32), whole.view(np.uint32))
