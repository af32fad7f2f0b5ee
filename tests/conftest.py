import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-context-reasoning_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


import pytest


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def tuning_lib():
    """The few tests that force a code path (MODCR_ATTN_DEBUG=8 = exact softmax pass, MODCR_GEMM_TN=1) run against
    libmodcr_hip_tuning.so: the product library has no environment knobs.  Everything else tests the product library."""
    import modcr_hip
    modcr_hip.use_tuning_library(True)
    try:
        yield modcr_hip
    finally:
        modcr_hip.use_tuning_library(False)
