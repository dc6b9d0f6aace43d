import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "amodal-depth-anything_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI binding; GPU tests fail (not skip) when the library or the device is missing."""
    import torch
    import hip_ext
    hip_ext.load()
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    return hip_ext


@pytest.fixture
def forced_tile(hip):
    """force(cfg, variant): pins ada_igemm's tile configuration / main-loop variant for the test (debug hooks of the library), restored afterwards."""
    def force(cfg, variant):
        hip.debug_set_tile(cfg)
        hip.debug_set_variant(variant)
    yield force
    hip.debug_set_tile(-1)
    hip.debug_set_variant(0)
    hip.debug_set_group(0)
