import importlib
import os
import sys

import numpy as np
import pytest

# torch first: it ships its own copy of the HIP runtime, and a process that lets librpt_hip.so bring in the system's libamdhip64
# BEFORE torch is imported leaves torch unable to see the GPU ("No HIP GPUs are available" at its first CUDA call) — whether a
# test that needs both worked used to depend on which test happened to import torch first.  bench.py has the same order.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rpt():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("rust-path-tracer_amd")


@pytest.fixture(scope="session")
def hipmod(rpt):
    return importlib.import_module("rust-path-tracer_amd.hip")


@pytest.fixture(scope="session")
def tiles(rpt):
    return importlib.import_module("rust-path-tracer_amd.tiles")


@pytest.fixture(scope="session")
def oracle():
    from oracle_ffi import Oracle
    return Oracle("rpt_math")


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle_ffi import Oracle
    return Oracle("libm")


_worlds = {}


@pytest.fixture(scope="session")
def world(rpt):
    def load(name):
        if name not in _worlds:
            _worlds[name] = rpt.World.from_path(rpt.fixture(name + ".glb"))
        return _worlds[name]
    return load


@pytest.fixture(scope="session")
def renderer(hipmod):
    """One Renderer for the whole GPU session (rpt_create is exercised once)."""
    r = hipmod.Renderer(0)
    yield r
    r.close()


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return float(d / n) if n > 0 else float(d)
