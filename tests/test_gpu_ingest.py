"""GPU parity on scenes that come out of the loader's texture / skybox paths: a GLB with embedded
base-colour, metallic-roughness and normal textures -> atlas -> k_shade<.., TEXTURED>, and a skybox FILE through the host
dispatch mirror (rpt_trace_gpu(scene, skybox_path, ..) = trace_gpu, src/trace.rs:136-224) — both against the oracle."""
import numpy as np
import pytest

from scenes import png_bytes, write_glb

pytestmark = pytest.mark.gpu


def _textured_glb(tmp_path):
    y, x = np.mgrid[0:64, 0:64]
    albedo = np.stack([(x * 4) % 256, (y * 4) % 256, ((x + y) * 2) % 256], -1).astype(np.uint8)
    mr = np.stack([(x * 3 + 40) % 256, (y * 2 + 90) % 256, np.full_like(x, 30)], -1).astype(np.uint8)
    nrm = np.stack([128 + 40 * np.sin(x / 5.0), 128 + 40 * np.cos(y / 7.0), np.full(x.shape, 230.0)], -1).astype(np.uint8)
    pos = np.array([[-3, 0, 6], [3, 0, 6], [3, 0, -2], [-3, 0, -2],          # glTF y-up floor ...
                    [-3, 0, 6], [3, 0, 6], [3, 4, 6], [-3, 4, 6]], np.float32)  # ... and a back wall
    nor = np.array([[0, 1, 0]] * 4 + [[0, 0, -1]] * 4, np.float32)
    uv = np.array([[0, 0], [2.5, 0], [2.5, 3], [0, 3], [0, 0], [1, 0], [1, 1], [0, 1]], np.float32)   # the floor wraps
    idx = np.array([0, 1, 2, 0, 2, 3, 4, 6, 5, 4, 7, 6], np.uint32)
    mats = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1}},
             "normalTexture": {"index": 2}}]
    return write_glb(str(tmp_path / "textured.glb"), pos, idx, normals=nor, uvs=uv, materials=mats,
                     images=[png_bytes(albedo), png_bytes(mr), png_bytes(nrm)])


def test_textured_glb_through_the_loader_matches_the_oracle(renderer, oracle, rpt, tmp_path):
    w = rpt.World.from_path(_textured_glb(tmp_path))
    assert w.atlas is not None and w.materials[0]["has_normal_texture"] == 1
    W, H, spp = 144, 96, 4
    cfg = rpt.default_config(W, H, cam_position=(0.0, 1.5, -4.0, 0.0))
    seeds = rpt.blue_noise_seeds(W, H)
    renderer.upload_scene(w)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)
    acc, _ = renderer.read_accum()
    ref, _, st = oracle.trace_cpu(cfg, oracle.scene(w), seeds, spp)
    g = renderer.stats()
    assert st.error_flags == 0 and g["extension_rays"] == st.extension_rays and g["sky_evals"] == st.sky_evals
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    assert ref[..., :3].std() > 0.02


def test_host_dispatch_with_a_skybox_file(oracle, rpt, world, tmp_path):
    """trace_gpu(scene_path, Some(skybox_path), state): the file is loaded as the reference's CPU path loads it (8-bit
    quantised, asset.rs:266-273) and used by the image-skybox branch (lib.rs:70-78); the framebuffer (mean) equals the
    oracle's on the same buffers bit for bit.  A skybox that cannot be read falls back to the 2x2 magenta image."""
    y, x = np.mgrid[0:32, 0:64]
    sky = np.stack([(x * 4) % 256, (y * 8) % 256, ((x * y) // 4) % 256], -1).astype(np.uint8)
    sky_path = tmp_path / "sky.png"
    sky_path.write_bytes(png_bytes(sky))
    W, H, spp = 96, 64, 8
    for path in (str(sky_path), str(tmp_path / "missing.png")):
        state = rpt.setup_trace(W, H, spp)
        state.config.has_skybox = 1
        state.set_sync_rate(spp)
        rpt.trace_gpu(rpt.fixture("PBRTest.glb"), path, state)
        frame = state.framebuffer()
        cfg = rpt.default_config(W, H, has_skybox=1)
        skybox = rpt.load_skybox(path) if path == str(sky_path) else None
        ref, _, _ = oracle.trace_cpu(cfg, oracle.scene(world("PBRTest"), skybox_f32=skybox), rpt.blue_noise_seeds(W, H), spp)
        want = ref[..., :3] / np.float32(spp)
        assert state.samples == spp
        assert np.array_equal(frame.view(np.uint32), want.view(np.uint32))
        state.close()


@pytest.mark.parametrize("sync_rate,target", [(3, 8), (4, 8), (32, 5), (1, 3)])
def test_host_dispatch_with_overlapped_read_back(oracle, rpt, world, sync_rate, target):
    """rpt_tracing_state_set_overlap: batch k+1 is enqueued before the image after batch k is read; the framebuffer the call
    ends with, and the sample count, are exactly those of the sequential loop — whatever the batch split."""
    W, H = 88, 60
    state = rpt.setup_trace(W, H, target)
    state.config.nee = 1
    state.set_sync_rate(sync_rate)
    state.set_overlap()
    rpt.trace_gpu(rpt.fixture("DarkCornell.glb"), None, state)
    cfg = rpt.default_config(W, H, nee=1)
    ref, _, _ = oracle.trace_cpu(cfg, oracle.scene(world("DarkCornell")), rpt.blue_noise_seeds(W, H), target)
    assert state.samples == target
    assert np.array_equal(state.framebuffer().view(np.uint32), (ref[..., :3] / np.float32(target)).view(np.uint32))
    state.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_host_dispatch_flush_from_another_thread(oracle, rpt, world, overlap):
    """The reference's interaction path (src/trace.rs:216-222): the render thread runs until told to stop; the UI thread
    writes a new configuration and raises `dirty`; the loop re-reads the configuration, zeroes the accumulators, restarts
    the sample count — and what it ends with is exactly the new view's image for the samples rendered since."""
    import threading
    import time
    W, H = 96, 64
    state = rpt.TracingState(rpt.host.lib().rpt_tracing_state_new(W, H))
    state.config.nee = 1
    state.set_sync_rate(2)
    state.set_overlap(overlap)
    state.set_running(True)
    errors = []

    def run():
        try:
            rpt.trace_gpu(rpt.fixture("DarkCornell.glb"), None, state)
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=run)
    t.start()
    deadline = time.time() + 120
    while state.samples < 6 and time.time() < deadline and t.is_alive():
        time.sleep(0.001)
    moved = rpt.default_config(W, H, nee=1, cam_position=(0.6, 1.4, -4.0, 0.0), cam_rotation=(0.05, -0.15, 0.0, 0.0))
    state.set_config(moved)
    state.set_dirty()
    seen_restart = False
    last = state.samples
    while time.time() < deadline and t.is_alive():
        s = state.samples
        if s < last:
            seen_restart = True
        last = s
        if seen_restart and s >= 6:
            break
        time.sleep(0.0005)
    state.set_running(False)
    t.join(120)
    assert not t.is_alive() and not errors, errors
    assert seen_restart
    n = state.samples
    frame = state.framebuffer()
    ref, _, _ = oracle.trace_cpu(moved, oracle.scene(world("DarkCornell")), rpt.blue_noise_seeds(W, H), n)
    assert n >= 2 and np.array_equal(frame.view(np.uint32), (ref[..., :3] / np.float32(n)).view(np.uint32))
    state.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_host_dispatch_keeps_publishing_while_interacting(oracle, rpt, world, overlap):
    """A camera drag holds `interacting` for many iterations (src/app.rs): with the flag already up the reference's inner loop ends after
    its FIRST sample (src/trace.rs:181-189), the 1-sample image is PUBLISHED (198-213 run before the flush of 216-222) and discarded.
    The framebuffer therefore follows the camera while the drag lasts, one sample per frame whatever sync_rate is — in the overlapped
    loop too, where a flushing iteration must read what it has just enqueued before the reset throws it away."""
    import threading
    import time
    W, H, rate = 96, 64, 3
    state = rpt.TracingState(rpt.host.lib().rpt_tracing_state_new(W, H))
    state.config.nee = 1
    state.set_sync_rate(rate)
    state.set_overlap(overlap)
    state.set_interacting(True)                                 # the drag starts before the first batch
    state.set_running(True)
    errors = []

    def run():
        try:
            rpt.trace_gpu(rpt.fixture("DarkCornell.glb"), None, state)
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=run)
    t.start()
    sc = oracle.scene(world("DarkCornell"))
    seeds = rpt.blue_noise_seeds(W, H)
    deadline = time.time() + 120
    views = [rpt.default_config(W, H, nee=1),
             rpt.default_config(W, H, nee=1, cam_position=(0.6, 1.4, -4.0, 0.0), cam_rotation=(0.05, -0.15, 0.0, 0.0)),
             rpt.default_config(W, H, nee=1, cam_position=(-0.4, 1.1, -3.5, 0.0), cam_rotation=(0.0, 0.2, 0.0, 0.0))]
    shown = []
    for k, view in enumerate(views):
        if k:
            state.set_config(view)                              # (picked up by the very next iteration: interacting flushes each one)
        ref, _, _ = oracle.trace_cpu(view, sc, seeds, 1)
        want = ref[..., :3].view(np.uint32)                       # (sum of one sample / 1.0)
        ok = False
        while time.time() < deadline and t.is_alive() and not ok:
            ok = np.array_equal(state.framebuffer().view(np.uint32), want)     # the drag's current view, ONE sample, from zero
            time.sleep(0.001)
        shown.append(ok)
    state.set_interacting(False)
    state.set_running(False)
    t.join(120)
    assert not t.is_alive() and not errors, errors
    assert shown == [True, True, True], shown
    state.close()


def test_host_dispatch_refuses_a_resize_while_rendering(rpt):
    """Every buffer of a trace_gpu call is sized for the resolution it started with (src/trace.rs:146-148): a configuration
    with another width / height written while it runs ends the call with an error instead of overrunning them."""
    import threading
    import time
    W, H = 64, 48
    state = rpt.TracingState(rpt.host.lib().rpt_tracing_state_new(W, H))
    state.set_sync_rate(2)
    state.set_running(True)
    result = []

    def run():
        try:
            rpt.trace_gpu(rpt.fixture("DarkCornell.glb"), None, state)
            result.append("returned")
        except rpt.host.HostError as e:
            result.append(str(e))

    t = threading.Thread(target=run)
    t.start()
    deadline = time.time() + 120
    while state.samples < 4 and time.time() < deadline and t.is_alive():
        time.sleep(0.001)
    state.set_config(rpt.default_config(W * 2, H))
    state.set_dirty()
    t.join(120)
    state.set_running(False)
    assert not t.is_alive() and result and "resolution changed" in result[0], result
    state.close()
