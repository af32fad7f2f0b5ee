import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-context-reasoning_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the library reads its tuning / debug knobs once per process unless this is set: tests that force a code path
# (e.g. MODCR_ATTN_DEBUG=8 = exact softmax pass) need them re-read per call
os.environ.setdefault("MODCR_ATTN_AB", "1")
os.environ.setdefault("MODCR_GEMM_AB", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
