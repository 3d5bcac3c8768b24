import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import refimport
    if refimport.available():
        return
    skip = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "reference" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _forced_gemm_kernel():
    """LAVIE_FORCE_TILE=<mode> runs the GPU tests with one GEMM kernel forced everywhere it is eligible
    (lavie_debug_force_tile; e.g. 7 = the persistent ping-pong kernel): kernel choice must never change results."""
    mode = os.environ.get("LAVIE_FORCE_TILE")
    if mode:
        from lavie_amd import _lib
        _lib.load().lavie_debug_force_tile(int(mode, 0))
    yield
