"""hijiki_amd — MI355X-native wavefront path tracer for the Hijiki hot path.

Layers:
  include/hijiki_hip.h   C ABI of the device path (HIP kernels for gfx950)
  include/hijiki_host.h  C ABI of the host-side scene model / compiler (C++)
  hijiki_amd.host        ctypes mirror of the Scene / Shape / Material API
  hijiki_amd.device      ctypes mirror of the render API (needs the HIP .so)
"""
from . import abi  # noqa: F401
from .host import Scene, CompiledScene, make_blocks, blocks_per_pass  # noqa: F401

__version__ = "0.1.0"
