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


@pytest.fixture(scope="session")
def ctx_exact():
    """Same library, AB_STACK_EXACT=1: the direct re-summing clipping engine (bit-exact cross-check)."""
    import astroburst_amd as ab
    old = os.environ.get("AB_STACK_EXACT")
    os.environ["AB_STACK_EXACT"] = "1"
    try:
        c = ab.Context(0)
    finally:
        if old is None:
            os.environ.pop("AB_STACK_EXACT", None)
        else:
            os.environ["AB_STACK_EXACT"] = old
    yield c
    c.close()
