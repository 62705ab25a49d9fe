import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pbrt-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One context per test session; fails loudly if the HIP library or the GPU is missing."""
    import pbrt_hip
    ctx = pbrt_hip.Context(0)
    yield ctx
    ctx.close()
