"""Host-side dispatch mirror (librpt_host.so): World::from_path, defaults, error behaviour — no GPU needed."""
import os

import numpy as np
import pytest


def test_default_config_matches_reference(rpt):
    """TracingConfig::default (shared_structs/src/lib.rs:27-42)."""
    c = rpt.default_config()
    assert (c.width, c.height, c.min_bounces, c.max_bounces, c.nee, c.has_skybox) == (1280, 720, 3, 4, 0, 0)
    assert list(c.cam_position) == [0.0, 1.0, -5.0, 0.0] and list(c.cam_rotation) == [0.0] * 4
    s = np.array(list(c.sun_direction), np.float32)
    v = np.array([0.5, 1.3, 1.0], np.float32)
    assert np.allclose(s[:3], v / np.linalg.norm(v), atol=1e-7) and s[3] == 15.0
    assert np.allclose(list(c.specular_weight_clamp), [0.1, 0.9])


def test_world_load_errors_are_reported_not_fatal(rpt, tmp_path):
    with pytest.raises(rpt.host.HostError):
        rpt.World.from_path(str(tmp_path / "missing.glb"))
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"not a glb file at all........")
    with pytest.raises(rpt.host.HostError):
        rpt.World.from_path(str(bad))


def test_axis_swap_and_winding(rpt, world):
    """asset.rs:102,106: positions are (x, z, y); DarkCornell's floor must end up at constant world Y."""
    w = world("DarkCornell")
    v = w.per_vertex["vertex"]
    assert np.all(v[:, 3] == 1.0) and np.all(w.per_vertex["normal"][:, 3] == 0.0)
    n = w.per_vertex["normal"][:, :3]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    # the default camera at (0, 1, -5) looking down +Z must see geometry in front of it
    assert v[:, 2].max() > 0 and v[:, 1].min() < 1.0 < v[:, 1].max()
    # emissive triangles face the scene: geometric normal (post winding swap) agrees with the shading normal
    tri = w.indices
    em = np.where(w.materials["emissive"][tri["material"]][:, :3].any(axis=1))[0]
    a, b, c = v[tri["v0"][em], :3], v[tri["v1"][em], :3], v[tri["v2"][em], :3]
    g = np.cross(b - a, c - a)
    assert np.all((g * n[tri["v0"][em]]).sum(axis=1) != 0)


def test_world_from_buffers_runs_reference_pipeline(rpt):
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 2, 2], [3, 2, 2], [2, 3, 2]], np.float32)
    tris = np.array([[0, 1, 2, 0], [0, 1, 3, 0], [4, 5, 6, 1]], np.uint32)
    mats = np.zeros(2, rpt._ffi.MATERIAL_DTYPE)
    mats["albedo"] = 1.0
    mats["emissive"][1] = [3, 3, 3, 0]
    w = rpt.World.from_buffers(verts, None, None, tris, mats)
    assert len(w.indices) == 3 and w.n_emissive_triangles == 1 and len(w.light_pick) == 1
    assert w.light_pick["ratio"][0] == 1.0 and w.light_pick["triangle_pick_pdf_a"][0] == 1.0
    assert np.isclose(w.light_pick["triangle_area_a"][0], 0.5, rtol=1e-6)
    leaves = w.nodes[w.nodes["triangle_count"] > 0]
    assert leaves["triangle_count"].sum() == 3


def test_trace_gpu_without_device_reports_error(rpt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    state = rpt.setup_trace(32, 32, 1)
    with pytest.raises(rpt.host.HostError) as e:
        rpt.trace_gpu(rpt.fixture("DarkCornell.glb"), None, state)
    assert "no HIP device" in str(e.value)
    assert state.samples == 0
    state.close()


def test_rptscene_cache_roundtrip_is_byte_identical(rpt, world, tmp_path):
    """the five POD buffers survive save/load bit for bit (Python writer and C loader agree)."""
    for name in ("DarkCornell", "PBRTest"):
        w = world(name)
        path = str(tmp_path / (name + ".rptscene"))
        w.save(path)
        w2 = rpt.World.from_cache(path)
        for k in ("per_vertex", "indices", "nodes", "materials", "light_pick"):
            assert np.array_equal(getattr(w, k).view(np.uint8), getattr(w2, k).view(np.uint8)), k
        assert w2.bvh_max_depth == w.bvh_max_depth and w2.n_emissive_triangles == w.n_emissive_triangles
    bad = tmp_path / "bad.rptscene"
    bad.write_bytes(b"RPTSCN01" + b"\x00" * 20)
    with pytest.raises(rpt.host.HostError):
        rpt.World.from_cache(str(bad))


def test_png_writer_srgb(rpt, tmp_path):
    from PIL import Image
    img = np.zeros((4, 5, 3), np.float32)
    img[0, 0] = [0.0, 0.5, 1.0]
    img[1, 1] = [0.0031308, 2.0, -1.0]
    img[2, 2] = [np.nan, 0.2140, 0.0]
    path = str(tmp_path / "a.png")
    rpt.host.write_png(path, img)
    px = np.array(Image.open(path))
    assert px.shape == (4, 5, 4) and np.all(px[..., 3] == 255)
    assert list(px[0, 0, :3]) == [0, 188, 255]          # 0.5 linear -> 0.7354 sRGB -> 188
    assert list(px[1, 1, :3]) == [10, 255, 0]           # 0.0031308 * 12.92 * 255 = 10.3 ; clamps
    assert list(px[2, 2, :3]) == [0, 127, 0]            # NaN -> 0 ; 0.214 -> ~0.5
    rpt.host.write_png(path, img, srgb=False)
    assert list(np.array(Image.open(path))[0, 0, :3]) == [0, 128, 255]


def _quad():
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    return pos, idx


def test_crafted_glb_is_rejected_without_reading_out_of_bounds(rpt, tmp_path):
    """A .glb is untrusted input: vertex indices beyond the primitive, accessor counts / offsets / strides that point
    outside the binary chunk or wrap around must be reported as load errors (the Rust reference panics safely)."""
    from scenes import write_glb
    pos, idx = _quad()
    ok = rpt.World.from_path(write_glb(str(tmp_path / "ok.glb"), pos, idx))
    assert len(ok.indices) == 2 and len(ok.per_vertex) == 4

    bad_idx = idx.copy()
    bad_idx[4] = 4                                        # one past the last vertex
    with pytest.raises(rpt.host.HostError):
        rpt.World.from_path(write_glb(str(tmp_path / "idx.glb"), pos, bad_idx))
    bad_idx[4] = 0xFFFFFFF0
    with pytest.raises(rpt.host.HostError):
        rpt.World.from_path(write_glb(str(tmp_path / "idx2.glb"), pos, bad_idx))

    def huge_count(acc, views):
        acc[0]["count"] = 1 << 40                         # would be a 24 TB resize
    def negative_count(acc, views):
        acc[-1]["count"] = -3
    def offset_past_end(acc, views):
        acc[0]["byteOffset"] = 1 << 50
    def wrapping_stride(acc, views):
        views[0]["byteStride"] = 1 << 62
    def fractional(acc, views):
        acc[-1]["count"] = 4.5
    for k, patch in enumerate((huge_count, negative_count, offset_past_end, wrapping_stride, fractional)):
        with pytest.raises(rpt.host.HostError):
            rpt.World.from_path(write_glb(str(tmp_path / f"acc{k}.glb"), pos, idx, accessor_patch=patch))


def test_scene_cache_with_a_cyclic_bvh_is_rejected_at_once(rpt, tmp_path):
    """An .rptscene file is untrusted input: a node array that is not a tree (here an inner node pointing back at the root —
    found as a HANG by tools/fuzz_glb.py) is an error, not an endless walk."""
    import time
    w = rpt.World.from_path(rpt.fixture("DarkCornell.glb"))
    good = tmp_path / "good.rptscene"
    w.save(str(good))
    data = bytearray(good.read_bytes())
    off = 8 + 6 * 8 + len(w.per_vertex) * 64 + len(w.indices) * 16           # the node array
    inner = int(np.nonzero(w.nodes["triangle_count"] == 0)[0][3])
    data[off + inner * 32 + 28: off + inner * 32 + 32] = (0).to_bytes(4, "little")     # left_or_first = 0: a cycle through the root
    bad = tmp_path / "bad.rptscene"
    bad.write_bytes(bytes(data))
    t = time.time()
    with pytest.raises(rpt.host.HostError):
        rpt.World.from_cache(str(bad))
    assert time.time() - t < 5.0
    assert len(rpt.World.from_cache(str(good)).nodes) == len(w.nodes)


def test_scene_cache_with_absurd_counts_is_rejected_before_anything_is_allocated(rpt, tmp_path):
    """The counts of an .rptscene header are untrusted: 2^31 - 1 vertices used to size a 128 GB vector before a single record was read
    (std::bad_alloc across the C ABI; tools/fuzz_glb.py under AddressSanitizer, round 4).  A header that promises more records than the
    file holds is an error at once."""
    import time
    w = rpt.World.from_path(rpt.fixture("DarkCornell.glb"))
    good = tmp_path / "good.rptscene"
    w.save(str(good))
    for field in range(5):                                                      # n_vertices, n_triangles, n_nodes, n_materials, n_light_pick
        data = bytearray(good.read_bytes())
        data[8 + 8 * field: 16 + 8 * field] = (0x7fffffff).to_bytes(8, "little")
        bad = tmp_path / f"bad{field}.rptscene"
        bad.write_bytes(bytes(data))
        t = time.time()
        with pytest.raises(rpt.host.HostError):
            rpt.World.from_cache(str(bad))
        assert time.time() - t < 2.0
    assert len(rpt.World.from_cache(str(good)).nodes) == len(w.nodes)


def test_loader_mutation_fuzz_slice():
    """A fixed-seed slice of tools/fuzz_glb.py (bit flips, truncations, splices, digit and length mutations of .glb, .obj,
    .png, .hdr and .rptscene files): every file is loaded or rejected, none crashes or hangs the loader."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_glb.py"), "160", "21"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "0 crashes" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def _soup(n, seed, poison=None):
    import importlib
    ffi = importlib.import_module("rust-path-tracer_amd._ffi")
    rng = np.random.default_rng(seed)
    grid = rng.integers(-3, 4, (n * 3, 3)).astype(np.float32) * 0.5
    grid[rng.random(grid.shape) < 0.15] = -0.0
    grid[rng.random(grid.shape) < 0.15] = 0.0
    if poison is not None:
        poison(grid, rng)
    v = np.concatenate([grid, np.ones((len(grid), 1), np.float32)], 1)
    t = np.zeros(n, ffi.TRIANGLE_DTYPE)
    idx = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
    nm = t.dtype.names
    t[nm[0]], t[nm[1]], t[nm[2]] = idx[:, 0], idx[:, 1], idx[:, 2]
    t[nm[3]] = rng.integers(0, 4, n)
    return v, t


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest", "PBRTest"])
def test_host_bvh_builder_equals_the_oracle_builder(rpt, oracle, world, scene):
    """The product's builder (csrc/host/bvh_build.cpp) against the oracle's statement-by-statement restatement of
    src/bvh.rs:59-324 (oracle/bvh_oracle.cpp, no shared code): same node pool — order, leaf ranges, bounds — and the same
    triangle order, byte for byte, from a shuffled input; and the World the loader produced holds exactly that tree for the
    file-order input (the builder runs inside World::from_path, src/asset.rs:196)."""
    w = world(scene)
    v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
    t = w.indices[np.random.default_rng(5).permutation(len(w.indices))]
    hn, ht = rpt.host.bvh_build(v, t)
    on, ot = oracle.bvh_build(v, t)
    assert hn.tobytes() == on.tobytes() and ht.tobytes() == ot.tobytes()
    # rebuilding from the World's own (already reordered) triangles is a fixed point only if the build is deterministic in its
    # input; what must hold is oracle(input) == host(input) for that input as well
    hn2, ht2 = rpt.host.bvh_build(v, w.indices)
    on2, ot2 = oracle.bvh_build(v, w.indices)
    assert hn2.tobytes() == on2.tobytes() and ht2.tobytes() == ot2.tobytes()


@pytest.mark.parametrize("bins", [2, 3, 16, 128])
def test_host_bvh_builder_equals_the_oracle_builder_on_tie_soups(rpt, oracle, bins):
    """Exactly-equal coordinates, +0 / -0, point triangles, 2..128 bins; infinities and a NaN coordinate (skipped by f32::min / max)."""
    v, t = _soup(3000, bins)
    hn, ht = rpt.host.bvh_build(v, t, bins)
    on, ot = oracle.bvh_build(v, t, bins)
    assert hn.tobytes() == on.tobytes() and ht.tobytes() == ot.tobytes()

    def hostile(g, r):
        g[r.random(g.shape) < 0.004] = np.inf
        g[r.random(g.shape) < 0.004] = -np.inf
        g[7, 1] = np.nan
    v, t = _soup(1500, 100 + bins, hostile)
    hn, ht = rpt.host.bvh_build(v, t, bins)
    on, ot = oracle.bvh_build(v, t, bins)
    assert hn.tobytes() == on.tobytes() and ht.tobytes() == ot.tobytes()


@pytest.mark.parametrize("scene", ["DarkCornell", "VeachMIS", "FurnaceTest", "PBRTest"])
def test_light_table_of_the_host_mirror_equals_the_oracles_restatement(oracle, world, scene):
    """src/light_pick.rs:13-122 restated twice — csrc/host/light_table.cpp (what World::from_path uses) and oracle/bvh_oracle.cpp
    oracle_light_table (the checker of the GPU build, tests/test_gpu_light_table.py): the same table bit for bit."""
    w = world(scene)
    v = np.ascontiguousarray(w.per_vertex["vertex"], np.float32).reshape(-1, 4)
    ref = oracle.light_table(v, w.indices, w.materials)
    assert np.array_equal(np.ascontiguousarray(w.light_pick).view(np.uint32).reshape(-1, 7), ref)
