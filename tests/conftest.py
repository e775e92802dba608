import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The in-tree libraries are built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()            # always: make is incremental, and a prebuilt .so that is older than its sources must not be tested
    yield
