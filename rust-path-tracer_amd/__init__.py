"""rust-path-tracer_amd — MI355X-native wavefront path tracer behind the reference's render dispatch.

Only what the hot path needs lives here:
  csrc/      HIP kernels + C ABI (librpt_hip.so) and the host-side dispatch mirror (librpt_host.so)
  host.py    World / seeds / setup_trace / trace_gpu  (ctypes over librpt_host.so)
  hip.py     Renderer: ctypes over the rpt.h C ABI (librpt_hip.so)
  tiles.py   framebuffer tile partition + gather for one-process-per-GPU runs
The CPU oracle is NOT part of this package (see oracle/, test infrastructure only).
"""
from . import _ffi  # noqa: F401
from .host import World, TracingState, blue_noise_seeds, default_config, fixture, load_skybox, setup_trace, trace_gpu  # noqa: F401
