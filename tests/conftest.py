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
    so = os.path.join(ROOT, "bhusie_amd", "libbhray.so")
    oso = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    exe = os.path.join(ROOT, "bhusie_amd", "bhray_render")
    if not (os.path.exists(so) and os.path.exists(oso) and os.path.exists(exe)):
        g.build()
    yield
