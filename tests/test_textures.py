"""Scene-ingestion fidelity, CPU only: embedded glTF textures -> the reference's texture atlas
(src/atlas.rs:26-90, src/asset.rs:135-192), KHR_materials_emissive_strength (opt-in), skybox files as the CPU path sees
them (src/asset.rs:238-273)."""
import struct

import os

import numpy as np
import pytest

from scenes import pack_rects, png_bytes, write_glb


def _quad_with_uv():
    pos = np.array([[-1, 0, 0], [1, 0, 0], [1, 2, 0], [-1, 2, 0]], np.float32)
    nor = np.tile(np.array([[0, 0, -1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    return pos, np.array([0, 1, 2, 0, 2, 3], np.uint32), nor, uv


def _pattern(n, seed, channels=3):
    """A cheap-to-compress but position-dependent image."""
    y, x = np.mgrid[0:n, 0:n]
    img = np.stack([(x * 7 + y * 3 + seed * 11) % 256, (x // 3 + 2 * y + seed) % 256, (x ^ y) % 256], -1).astype(np.uint8)
    return img if channels == 3 else np.concatenate([img, np.full((n, n, 1), 200, np.uint8)], -1)


def test_atlas_layout_matches_the_reference_packer(rpt, tmp_path):
    """One material with base colour (2048^2), metallic-roughness (64^2) and normal (1024^2) textures = 4 atlas entries
    (assimp lists the metallic-roughness image under two texture types, so the reference atlases it twice): rectangles,
    uvst (with the reference's y / atlas_WIDTH), vertical flip, gamma -> linear of the albedo only, flags."""
    pos, idx, nor, uv = _quad_with_uv()
    albedo, mr, normal = _pattern(2048, 1), _pattern(64, 2), _pattern(1024, 3, channels=4)
    mats = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1},
                                      "baseColorFactor": [0.5, 0.6, 0.7, 1.0]}, "normalTexture": {"index": 2}}]
    path = write_glb(str(tmp_path / "tex.glb"), pos, idx, normals=nor, uvs=uv, materials=mats,
                     images=[png_bytes(albedo), png_bytes(mr), png_bytes(normal)])
    w = rpt.World.from_path(path)
    m = w.materials[0]
    assert (m["has_albedo_texture"], m["has_metallic_texture"], m["has_roughness_texture"], m["has_normal_texture"]) == (1, 1, 1, 1)
    assert w.atlas is not None and w.atlas.shape == (4096, 4096, 4)
    rects = pack_rects(4)
    assert [r[2] for r in rects] == [2048, 2048, 2048, 1024]
    for field, (x, y, rw, rh) in zip(("albedo", "metallic", "roughness", "normals"), rects):
        want = np.array([x / 4096, y / 4096, rw / 4096, rh / 4096], np.float32)
        assert np.array_equal(m[field], want), field
    # albedo: exact copy (texture size == leaf size), gamma -> linear with truncation, alpha 255, rows flipped
    lut = np.array([int(np.float32(np.float32(v / np.float32(255.0)) ** np.float32(2.2)) * np.float32(255.0)) for v in range(256)], np.uint8)
    x, y, rw, rh = rects[0]
    got = w.atlas[y:y + rh, x:x + rw]
    want = lut[albedo][::-1]
    close = np.abs(got[..., :3].astype(int) - want.astype(int))
    assert close.max() <= 1 and (close != 0).mean() < 1e-3      # powf of this libm vs the correctly rounded one: the odd boundary value
    assert np.all(got[..., 3] == 255)
    # normal map: exact copy into its 1024^2 leaf, alpha kept (to_rgba8), flipped
    x, y, rw, rh = rects[3]
    assert np.array_equal(w.atlas[y:y + rh, x:x + rw], normal[::-1])
    # metallic-roughness: the same image in two leaves, upscaled 64 -> 2048 (Lanczos3): identical copies, in range, smooth
    (x1, y1, s1, _), (x2, y2, s2, _) = rects[1], rects[2]
    a, b = w.atlas[y1:y1 + s1, x1:x1 + s1], w.atlas[y2:y2 + s2, x2:x2 + s2]
    assert np.array_equal(a, b)
    centre = a[::-1][16::32, 16::32, :3].astype(int)            # the texel centres of the source land on every 32nd output pixel
    assert np.abs(centre - mr.astype(int)).mean() < 12
    # everything outside the leaves stays zero
    mask = np.ones((4096, 4096), bool)
    for x, y, rw, rh in rects:
        mask[y:y + rh, x:x + rw] = False
    assert not w.atlas[mask].any()
    # tangents were derived (no TANGENT attribute): unit length, orthogonal to the normal
    t, n = w.per_vertex["tangent"][:, :3], w.per_vertex["normal"][:, :3]
    assert np.allclose(np.linalg.norm(t, axis=1), 1.0, atol=1e-5) and np.allclose((t * n).sum(1), 0.0, atol=1e-5)


def test_untextured_scenes_keep_no_atlas_and_shipped_scenes_are_unchanged(rpt, world):
    for name in ("DarkCornell", "PBRTest"):
        w = world(name)
        assert w.atlas is None
        assert not w.materials["has_albedo_texture"].any() and not w.materials["has_normal_texture"].any()


def test_emissive_strength_is_opt_in(rpt, tmp_path):
    pos, idx, nor, uv = _quad_with_uv()
    mats = [{"emissiveFactor": [1.0, 0.5, 0.25], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}}]
    path = write_glb(str(tmp_path / "em.glb"), pos, idx, normals=nor, uvs=uv, materials=mats)
    ref = rpt.World.from_path(path)                               # the reference ignores the extension: x 15 (asset.rs:163-166)
    assert np.array_equal(ref.materials[0]["emissive"], np.array([15.0, 7.5, 3.75, 15.0], np.float32))
    ext = rpt.World.from_path(path, emissive_strength=True)
    assert np.array_equal(ext.materials[0]["emissive"], np.array([4.0, 2.0, 1.0, 4.0], np.float32))
    assert ext.n_emissive_triangles == ref.n_emissive_triangles == 2


def _write_hdr(path, rgbe, rle):
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
        for row in rgbe:
            if not rle:
                f.write(row.tobytes())
                continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                comp = row[:, c]
                x = 0
                while x < w:                                      # literal runs of up to 100 bytes, and a run where values repeat
                    run = 1
                    while x + run < w and run < 127 and comp[x + run] == comp[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(comp[x])]))
                        x += run
                    else:
                        n = min(100, w - x)
                        f.write(bytes([n]) + comp[x:x + n].tobytes())
                        x += n


@pytest.mark.parametrize("rle", [False, True])
def test_hdr_skybox_is_quantised_like_the_cpu_path(rpt, tmp_path, rle):
    """dynamic_image_to_cpu_buffer (asset.rs:266-273): whatever range the file holds, the CPU path keeps
    round(clamp(v, 0, 1) * 255) / 255 per channel and alpha 1."""
    rng = np.random.default_rng(4)
    h, w = 12, 40
    rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[..., 3] = rng.integers(120, 132, (h, w))               # values from ~1e-5 to ~8
    rgbe[0, :8, 3] = 0                                           # e == 0: black
    rgbe[1, :, 0] = 77                                           # a long run for the RLE writer
    path = str(tmp_path / "sky.hdr")
    _write_hdr(path, rgbe, rle)
    got = rpt.load_skybox(path)
    scale = np.where(rgbe[..., 3:] == 0, 0.0, np.exp2(rgbe[..., 3:].astype(np.float64) - 136.0)).astype(np.float32)
    lin = rgbe[..., :3].astype(np.float32) * scale
    q = np.floor(np.clip(lin, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.float32) / np.float32(255.0)
    assert got.shape == (h, w, 4) and np.all(got[..., 3] == 1.0)
    assert np.array_equal(got[..., :3], q)
    assert len(np.unique(got[..., :3])) <= 256


def test_png_skybox_and_decoder_variants(rpt, tmp_path):
    img = _pattern(48, 9, channels=4)
    p = tmp_path / "sky.png"
    p.write_bytes(png_bytes(img))
    got = rpt.load_skybox(str(p))
    assert np.array_equal(got[..., :3], img[..., :3].astype(np.float32) / np.float32(255.0)) and np.all(got[..., 3] == 1.0)
    with pytest.raises(rpt.host.HostError):
        rpt.load_skybox(str(tmp_path / "missing.png"))
    bad = tmp_path / "bad.hdr"
    bad.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n" + b"\x01" * 10)
    with pytest.raises(rpt.host.HostError):
        rpt.load_skybox(str(bad))


def test_obj_and_glb_of_the_same_mesh_give_the_same_world(rpt, tmp_path):
    """Wavefront OBJ + MTL through the same pipeline as GLB (asset.rs:78-128 treats every assimp import alike): (x, z, y)
    positions, (0, 2, 1) winding, Kd -> albedo, Ke x 15 -> emissive, Pm / Pr -> metallic / roughness; polygons fanned,
    negative indices, smooth normals where the file has none."""
    (tmp_path / "m.mtl").write_text("newmtl wall\nKd 0.7 0.6 0.5\nPr 0.4\nPm 0.1\nnewmtl lamp\nKd 1 1 1\nKe 1.0 0.5 0.25\n")
    (tmp_path / "box.obj").write_text(
        "mtllib m.mtl\n"
        "v -1 0 0\nv 1 0 0\nv 1 2 0\nv -1 2 0\nv -1 3 0\nv -1 3 1\nv 1 3 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 -1\n"
        "usemtl wall\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"          # a quad: fanned into two triangles
        "usemtl lamp\nf -3 -2 -1\n")                        # negative indices, no vt / vn: smooth normal generated
    w = rpt.World.from_path(str(tmp_path / "box.obj"))
    assert len(w.indices) == 3 and len(w.materials) == 2 and w.n_emissive_triangles == 1
    m = w.materials
    assert np.allclose(m[0]["albedo"], [0.7, 0.6, 0.5, 1.0]) and np.all(m[0]["roughness"] == np.float32(0.4)) and np.all(m[0]["metallic"] == np.float32(0.1))
    assert np.array_equal(m[1]["emissive"], np.array([15.0, 7.5, 3.75, 15.0], np.float32))
    v = w.per_vertex["vertex"]
    assert len(v) == 7 and np.all(v[:, 3] == 1.0)                      # 4 joined quad corners + 3 lamp corners
    assert {tuple(p) for p in v[:4, :3].tolist()} == {(-1, 0, 0), (1, 0, 0), (1, 0, 2), (-1, 0, 2)}        # (x, z, y)
    assert np.allclose(w.per_vertex["normal"][:4, :3], [0, -1, 0])                                           # vn (0, 0, -1) swapped
    lamp = [t for t in w.indices if t["material"] == 1][0]
    n = w.per_vertex["normal"][[lamp["v0"], lamp["v1"], lamp["v2"]], :3]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6) and np.allclose(n, n[0])
    # the same geometry as GLB gives the same triangles (as sets of corner positions) and the same materials' colours
    from scenes import write_glb
    pos = np.array([[-1, 0, 0], [1, 0, 0], [1, 2, 0], [-1, 2, 0]], np.float32)
    g = rpt.World.from_path(write_glb(str(tmp_path / "quad.glb"), pos, np.array([0, 1, 2, 0, 2, 3], np.uint32),
                                      normals=np.tile(np.array([[0, 0, -1]], np.float32), (4, 1)),
                                      materials=[{"pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.6, 0.5, 1.0]}}]))

    def tri_set(world, material):
        vv = world.per_vertex["vertex"][:, :3]
        return {tuple(sorted(map(tuple, vv[[t["v0"], t["v1"], t["v2"]]].tolist()))) for t in world.indices if t["material"] == material}
    assert tri_set(w, 0) == tri_set(g, 0)
    with pytest.raises(rpt.host.HostError):
        bad = tmp_path / "bad.obj"
        bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
        rpt.World.from_path(str(bad))


def test_image_headers_cannot_size_gigabytes_and_obj_relative_indices_are_bounded(rpt, tmp_path):
    """Untrusted files: a PNG / .hdr header claiming a huge image is refused before anything is allocated from it, and an
    OBJ corner index that reaches before the first element of its array is an error, not 'no attribute'."""
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 16384, 16384, 16, 6, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    (tmp_path / "bomb.png").write_bytes(bomb)
    with pytest.raises(rpt.host.HostError):
        rpt.load_skybox(str(tmp_path / "bomb.png"))
    (tmp_path / "bomb.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 32768 +X 32768\n")
    with pytest.raises(rpt.host.HostError):
        rpt.load_skybox(str(tmp_path / "bomb.hdr"))
    for face in ("f 1/-9 2 3", "f 1//-2 2 3", "f -4 2 3"):
        (tmp_path / "rel.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n" + face + "\n")
        with pytest.raises(rpt.host.HostError):
            rpt.World.from_path(str(tmp_path / "rel.obj"))
    (tmp_path / "ok.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/-1/-1 2/1/1 3//1\n")
    assert len(rpt.World.from_path(str(tmp_path / "ok.obj")).indices) == 1
    # lines that END at the keyword ("vn", "vt", "v", also behind long leading blanks, which puts the line on the heap): the
    # scanner must not start one byte past the terminator; such lines carry no data and are skipped; "vnormal" is no keyword
    pad = " " * 300
    (tmp_path / "bare.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn\nvt\nv\n" + pad + "vn\n" + pad + "vt\n" + pad + "v\nvn\t0 0 1\nvt\t0 0\n"
                                       "vnormal 9 9 9\nf 1/1/1 2/1/1 3/1/1\nvn")
    wb = rpt.World.from_path(str(tmp_path / "bare.obj"))
    assert len(wb.indices) == 1 and np.allclose(wb.per_vertex["normal"][:, :3], [[0, 1, 0]] * 3)      # (0,0,1) after the (x,z,y) swap


JPEG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
# sha256[:16] of the RGBA8 this decoder produces for the committed files (regression), and what libjpeg-turbo (Pillow) gave
# for them when they were made (tests/golden/jpeg/*.libjpeg.npy): two conforming decoders differ by a couple of LSBs
JPEG_EXPECTED = {"baseline_420_rst.jpg": "acced68f27089f48", "baseline_444.jpg": "c313e3e5c1665414", "grey.jpg": "f1c64c873ac52b14",
                 "progressive_422.jpg": "9e342789d67d204f"}


def test_jpeg_decoder_on_committed_files(rpt):
    """JPEG textures / skyboxes (csrc/host/jpeg_decode.cpp, restating jpeg-decoder 0.3 = what the reference's `image` crate
    uses): baseline 4:2:0 with restart markers, baseline 4:4:4, progressive 4:2:2, greyscale — decoded bytes are stable, within
    3 LSB of libjpeg-turbo's decode of the same files (mean error < 0.25), alpha 255, grey replicated."""
    import hashlib
    for name, want in JPEG_EXPECTED.items():
        got = (rpt.load_skybox(os.path.join(JPEG_DIR, name)) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
        assert got.shape == (35, 45, 4) and np.all(got[..., 3] == 255)
        assert hashlib.sha256(got.tobytes()).hexdigest()[:16] == want, name
        ref = np.load(os.path.join(JPEG_DIR, name[:-4] + ".libjpeg.npy")).astype(np.int32)
        d = np.abs(got[..., :3].astype(np.int32) - ref)
        assert d.max() <= 3 and d.mean() < 0.25, (name, d.max(), d.mean())
        if name == "grey.jpg":
            assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])


def test_jpeg_decoder_against_pillow_over_encoder_settings(rpt, tmp_path):
    """Every combination a glTF exporter produces — quality 60..95, 4:4:4 / 4:2:2 / 4:2:0, baseline / progressive / optimised
    tables, restart intervals, greyscale, sizes that are not multiples of the MCU down to 1 x 1 — against Pillow's decoder."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)

    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0), x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1)], -1).astype(np.float64)
        img[h // 4:h // 2, w // 4:w // 2] = [250, 10, 30]
        return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)

    settings = [dict(quality=90, subsampling=0), dict(quality=75, subsampling=1), dict(quality=60, subsampling=2),
                dict(quality=85, subsampling=2, progressive=True), dict(quality=95, subsampling=0, progressive=True),
                dict(quality=80, subsampling=1, progressive=True, optimize=True), dict(quality=80, subsampling=2, restart_marker_blocks=3),
                dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=5)]
    n = 0
    for (w, h) in [(64, 48), (97, 61), (8, 8), (1, 1), (17, 3), (200, 133)]:
        for k, opts in enumerate(settings + [dict(quality=85)]):
            im = Image.fromarray(picture(w, h))
            if k == len(settings):
                im = im.convert("L")
            path = str(tmp_path / "t.jpg")
            try:
                im.save(path, "JPEG", **opts)
            except TypeError:
                continue                                        # an older Pillow without restart_marker_blocks
            ref = np.asarray(Image.open(path).convert("RGB")).astype(np.int32)
            got = (rpt.load_skybox(path)[..., :3] * np.float32(255.0) + np.float32(0.5)).astype(np.int32)
            d = np.abs(got - ref)
            assert got.shape == ref.shape and d.max() <= 4 and d.mean() < (0.3 if w * h >= 1000 else 0.6), (w, h, opts, d.max(), d.mean())
            n += 1
    assert n >= 40


def test_glb_with_a_jpeg_texture_loads_like_its_png_twin(rpt, tmp_path):
    """A GLB whose base-colour image is a JPEG (glTF allows PNG and JPEG) goes through the same atlas path as a PNG: same
    atlas geometry and material flags, texels within JPEG's own error of the lossless twin."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from scenes import png_bytes, write_glb
    yy, xx = np.mgrid[0:32, 0:32]
    rgb = np.stack([60 + 5 * xx, 200 - 4 * yy, 90 + 2 * xx + 2 * yy], -1).astype(np.uint8)      # smooth: JPEG at quality 95 is nearly lossless on it
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, "JPEG", quality=95, subsampling=0)
    pos = np.array([[-1, 0, 0], [1, 0, 0], [1, 2, 0], [-1, 2, 0]], np.float32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    mats = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    worlds = []
    for name, blob in (("p.glb", png_bytes(rgb)), ("j.glb", buf.getvalue())):
        path = write_glb(str(tmp_path / name), pos, np.array([0, 1, 2, 0, 2, 3], np.uint32), normals=np.tile(np.array([[0, 0, -1]], np.float32), (4, 1)),
                         uvs=uv, materials=mats, images=[blob])
        worlds.append(rpt.World.from_path(path))
    wp, wj = worlds
    assert wj.atlas is not None and wp.atlas.shape == wj.atlas.shape
    assert wp.materials.tobytes() == wj.materials.tobytes()
    d = np.abs(wp.atlas.astype(np.int32) - wj.atlas.astype(np.int32))
    assert d.max() <= 12 and d.mean() < 1.5                     # (albedo gamma -> linear amplifies JPEG's small error)
