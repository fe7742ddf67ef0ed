import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "backgammon-engine_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Always through build(): it is content-based (the library carries the digest of the sources it was compiled
    from) and a no-op when fresh, so the tests can never run a libbgamd.so that was not built from the tree they sit
    in (*.so is git-ignored but travels with gpurun snapshots).  hipcc cross-compiles gfx950 without a GPU."""
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def weights():
    import numpy as np
    w = np.fromfile(os.path.join(GOLDEN, "tdgammonNEW100k.f32"), dtype=np.float32)
    assert w.size == 25601
    return w
