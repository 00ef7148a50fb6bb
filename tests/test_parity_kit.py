"""The parity kit for the REAL reference (tests/golden/parity_kit, tools/export_parity_kit.py, ffi/parity_test.rs): inputs +
expected accumulators a maintainer with a Rust toolchain feeds to kernels::trace_pixel.  Here (no rustc): the committed kit is
exactly what the exporter produces today — byte for byte — its manifest is consistent, the reference's own furnace assertion
(tests/correctness_tests.rs:14-33) holds on the committed accumulators, and the Rust file names the files the kit holds."""
import hashlib
import importlib.util
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tests", "golden", "parity_kit")


def _exporter():
    spec = importlib.util.spec_from_file_location("export_parity_kit", os.path.join(ROOT, "tools", "export_parity_kit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_kit_is_what_the_exporter_writes(tmp_path):
    manifest = json.load(open(os.path.join(KIT, "manifest.json")))
    fresh = _exporter().export(str(tmp_path))
    assert fresh["cases"] == manifest["cases"]
    assert set(fresh["files"]) == set(manifest["files"]) == set(os.listdir(KIT)) - {"manifest.json"}
    for name, meta in manifest["files"].items():
        committed = open(os.path.join(KIT, name), "rb").read()
        assert len(committed) == meta["bytes"] and hashlib.sha256(committed).hexdigest() == meta["sha256"], name
        if name.endswith(".accum_libm.bin") and fresh["libm"] != manifest["libm"]:
            continue                                            # another glibc may round a transcendental differently: informational file
        assert open(tmp_path / name, "rb").read() == committed, name


def test_kit_accumulators_hold_the_references_furnace_assertion(rpt):
    manifest = json.load(open(os.path.join(KIT, "manifest.json")))
    seen = 0
    for case in manifest["cases"]:
        W, H, spp = case["width"], case["height"], case["spp"]
        acc = np.fromfile(os.path.join(KIT, case["accum"]), np.float32).reshape(H, W, 4)
        libm = np.fromfile(os.path.join(KIT, case["accum_libm"]), np.float32).reshape(H, W, 4)
        assert np.all(acc[..., 3] == spp) and np.isfinite(acc).all()
        # the two oracle builds (shared correctly rounded math vs glibc) agree far below the 1e-4 bar: <= 2e-9 on the closed scenes,
        # 1.1e-6 on VeachMIS (an open scene: 84 float-only exp per sky hit, and a last-bit difference in sin / cos / acos can flip a
        # lobe choice of a single sample)
        err = np.linalg.norm(acc[..., :3].astype(np.float64) - libm[..., :3]) / np.linalg.norm(libm[..., :3].astype(np.float64))
        assert err < (1e-5 if case["scene"].startswith("VeachMIS") else 1e-6), (case["name"], err)
        cfg = np.fromfile(os.path.join(KIT, case["config"]), np.uint8)
        assert cfg.size == 80 and bytes(cfg) == bytes(rpt.default_config(W, H, nee=case["nee"], **{k: tuple(v) for k, v in case["config_overrides"].items()}))
        seeds = np.fromfile(os.path.join(KIT, case["seeds"]), np.uint32).reshape(H, W, 2)
        assert not seeds[..., 0].any() and seeds[0, 0, 1] == 1448498816 and seeds[75 % 256, 65 % 256, 1] == 50529028
        if case["scene"].startswith("FurnaceTest"):
            px = (acc[75, 65, :3] / np.float32(spp)).astype(np.float64) ** (1 / 2.2)      # tests/correctness_tests.rs:26-31
            assert np.all(np.abs(px - 0.8) < 0.02), px
            seen += 1
        w = rpt.World.from_cache(os.path.join(KIT, case["scene"]))
        assert len(w.indices) > 0
    assert seen == 2


def test_rust_side_reads_the_files_the_kit_holds():
    text = open(os.path.join(ROOT, "ffi", "parity_test.rs")).read()
    manifest = json.load(open(os.path.join(KIT, "manifest.json")))
    for case in manifest["cases"]:
        m = re.search(r'run_case\("%s", "([A-Za-z]+\.rptscene)", (\d+), (\d+), (\d+)\)' % case["name"], text)
        assert m, case["name"]
        assert (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))) == (case["scene"], case["width"], case["height"], case["spp"])
    for suffix in (".config.bin", ".accum.bin", ".accum_libm.bin", "seeds_{width}x{height}.bin"):
        assert suffix in text
    assert "kernels::trace_pixel(" in text and "1e-4" in text
