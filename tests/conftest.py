import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_loader
    return oracle_loader.load()


@pytest.fixture(scope="session")
def plugin():
    """One GPU context for the whole session (gpu tests only)."""
    import bevyray_amd as brt
    p = brt.RaytracePlugin([0])
    yield p
    p.close()
