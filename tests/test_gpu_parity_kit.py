"""The HIP path against the COMMITTED bytes of tests/golden/parity_kit — no live oracle in this file.

Each case is rendered through the C ABI from exactly what the kit holds — the `.rptscene` buffers, the seed buffer, the 80
`TracingConfig` bytes — and the accumulators must equal `<case>.accum.bin` bit for bit; the ray counts must equal the manifest's.
(`tests/test_parity_kit.py` keeps those files equal to what the oracle produces today; this test makes a stale or edited `.bin`
a GPU failure instead of a silent one.)  The furnace cases are the reference's own `furnace_test_cpu{,_mis}` settings
(tests/correctness_tests.rs:14-53: 128 x 128, 32 spp, pixel (65, 75))."""
import ctypes as C
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tests", "golden", "parity_kit")
MANIFEST = json.load(open(os.path.join(KIT, "manifest.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("case", MANIFEST["cases"], ids=[c["name"] for c in MANIFEST["cases"]])
def test_gpu_image_equals_the_committed_accumulators(renderer, rpt, case):
    ffi = __import__("importlib").import_module("rust-path-tracer_amd._ffi")
    W, H, spp = case["width"], case["height"], case["spp"]
    world = rpt.World.from_cache(os.path.join(KIT, case["scene"]))
    raw = open(os.path.join(KIT, case["config"]), "rb").read()
    assert len(raw) == C.sizeof(ffi.TracingConfig) == 80
    cfg = ffi.TracingConfig.from_buffer_copy(raw)
    assert (cfg.width, cfg.height, cfg.nee) == (W, H, case["nee"])
    seeds = np.fromfile(os.path.join(KIT, case["seeds"]), ffi.RNG_DTYPE)
    assert seeds.size == W * H
    want = np.fromfile(os.path.join(KIT, case["accum"]), np.float32).reshape(H, W, 4)
    renderer.upload_scene(world)
    renderer.set_config(cfg)
    renderer.reset(seeds)
    renderer.render(spp)                         # one call, as the kit's accumulators were made (the sum order does not depend on batching)
    acc, samples = renderer.read_accum()
    st = renderer.stats()
    assert samples == spp
    assert st["extension_rays"] == case["extension_rays"] and st["shadow_rays"] == case["shadow_rays"]
    diff = int((acc.view(np.uint32) != want.view(np.uint32)).sum())
    assert diff == 0, f"{case['name']}: {diff} of {acc.size} words differ from the committed accumulators"
    if case["scene"].startswith("FurnaceTest"):
        px = (acc[75, 65, :3] / np.float32(spp)).astype(np.float64) ** (1 / 2.2)      # tests/correctness_tests.rs:26-31
        assert np.all(np.abs(px - 0.8) < 0.02), px
