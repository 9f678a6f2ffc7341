import os
import sys

import pytest

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing torch BEFORE
# libhijiki_hip.so is loaded makes the dynamic linker resolve the library's NEEDED libamdhip64.so.7 to the copy
# torch already mapped; the other order leaves two runtimes in the process and torch then sees no GPU.
import torch  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import hj_oracle
    hj_oracle.lib()
    return hj_oracle


@pytest.fixture(scope="session")
def cbox():
    """Compiled synthetic Cornell-box scene (6332 triangles)."""
    from hijiki_amd import host
    scene = host.Scene.synthetic(host.SYNTH_CBOX)
    return scene.compile()


@pytest.fixture(scope="session")
def cbox_small():
    """Cornell box with a 320-triangle object: fast enough for linear-scan comparisons."""
    from hijiki_amd import host
    scene = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320)
    return scene.compile()


@pytest.fixture(scope="session")
def cbox_spheres():
    from hijiki_amd import host
    return host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()


@pytest.fixture(scope="session")
def gpu_renderer():
    from hijiki_amd import device
    r = device.Renderer(0)   # raises (fails the test) when the HIP library or the GPU is missing
    yield r
    r.close()
