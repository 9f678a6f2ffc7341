"""hijiki_amd — MI355X-native wavefront path tracer for the Hijiki hot path.

Layers:
  include/hijiki_hip.h   C ABI of the device path (HIP kernels for gfx950)
  include/hijiki_host.h  C ABI of the host-side scene model / compiler (C++)
  hijiki_amd.host        ctypes mirror of the Scene / Shape / Material API
  hijiki_amd.device      ctypes mirror of the render API (needs the HIP .so)
"""
import os as _os

# The renderer keeps three batches in flight on their own HIP streams.  The HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise: in a process where
# PyTorch created its streams first, two of the batch streams aliased and a frame took 9 % longer (measured:
# 276 -> 253 ms on cbox 1024^2 x 512 spp).  The variable is read when the HIP runtime initialises, so it has to
# be in the environment before the first HIP call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import abi  # noqa: F401,E402
from .host import Scene, CompiledScene, make_blocks, blocks_per_pass  # noqa: F401,E402

__version__ = "0.1.0"
