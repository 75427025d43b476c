import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/liboracle.so) -- the checker, never the thing under test."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """The product under test: libastroburst_hip.so through astroburst_amd.Context (gfx950 only)."""
    import astroburst_amd as ab
    c = ab.Context(0)          # raises loudly without the .so or without an MI355X
    yield c
    c.close()
