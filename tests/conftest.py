import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the in-tree libraries.  hipcc cross-compiles without a GPU."""
    from necat_amd import build
    build.build_hip()
    build.build_xcheck()          # the same sources + the retired kernel families: the alternative-path cases load it (capi.needs_xcheck)
    from tests import oracle_build
    oracle_build.build_oracle()
    return build


@pytest.fixture(scope="session")
def ctx(built):
    from necat_amd import capi
    c = capi.Context(0)     # raises loudly when no GPU is usable - by design, no CPU fallback
    yield c
    c.close()
