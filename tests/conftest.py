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


@pytest.fixture(scope="session")
def ctx_deep():
    """Same library with AB_STACK_DEEP_FROM / AB_BATCH_DEEP_FROM = 64: every stack of more than 64 frames takes the
    workgroup-per-pixel kernels (stack_deep.hip, scms_deep_kernel) that the default dispatch reserves for > 4096 / > 2048 frames."""
    import astroburst_amd as ab
    old = {k: os.environ.get(k) for k in ("AB_STACK_DEEP_FROM", "AB_BATCH_DEEP_FROM")}
    os.environ.update(AB_STACK_DEEP_FROM="64", AB_BATCH_DEEP_FROM="64")
    try:
        c = ab.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield c
    c.close()


DEV_ONLY = ("needs the developer library: `make -C astroburst_amd/csrc dev` and AB_LIB_PATH=astroburst_amd/libastroburst_hip_dev.so "
            "(the release library compiles the superseded forms' switches, the sweep knobs and the fault injection out: ab_dev_env)")


def require_dev_build():
    """Skip unless the loaded library is the -DAB_DEV_ABLATION build: the switches a test is about to set do nothing in the release."""
    import astroburst_amd as ab
    if not ab.is_dev_build():
        pytest.skip(DEV_ONLY)


@pytest.fixture(scope="session")
def dev_build():
    require_dev_build()


def _ctx_under(env):
    """A context created under developer switches (dev build only)."""
    import astroburst_amd as ab
    require_dev_build()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return ab.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="session")
def ctx_midjoin():
    """Same library, AB_DETECT_MIDJOIN=1: the grouped detection with a host join after the root numbering (the default is one chain)."""
    c = _ctx_under({"AB_DETECT_MIDJOIN": "1"})
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_pixel_list():
    """Same library, AB_DETECT_NO_RECS=1: the chained detection over the tiles' PIXEL lists (roots + comp_stats) instead of one record
    per tile-local component (label_tile_body<true, true> + comp_merge)."""
    c = _ctx_under({"AB_DETECT_NO_RECS": "1"})
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_pixelwise():
    """Same library, AB_LABEL_PIXELWISE=1: the tile-local unions pixel by pixel instead of run by run (and hence the pixel-list chain)."""
    c = _ctx_under({"AB_LABEL_PIXELWISE": "1"})
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_r4_detect():
    """Same library with round 4's detection forms (AB_LABEL_LEGACY=1: two-pass labelling; AB_DETECT_FULL_RECORDS=1: every component's
    record crosses to the host): the cross-check of the tile-local union-find and of the device-side selection of the brightest."""
    import astroburst_amd as ab
    require_dev_build()
    old = {k: os.environ.get(k) for k in ("AB_LABEL_LEGACY", "AB_DETECT_FULL_RECORDS")}
    os.environ.update(AB_LABEL_LEGACY="1", AB_DETECT_FULL_RECORDS="1")
    try:
        c = ab.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield c
    c.close()
