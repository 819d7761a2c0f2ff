import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _eager_collector():
    """FLM_EAGER_GC=1 makes Python's cyclic collector run every few allocations, the timing under which a test
    that reads an allocator counter across a call shows whether it depends on when earlier tests' tensors die."""
    if os.environ.get("FLM_EAGER_GC") == "1":
        import gc
        old = gc.get_threshold()
        gc.set_threshold(50, 5, 5)
        yield
        gc.set_threshold(*old)
    else:
        yield
