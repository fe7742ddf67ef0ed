import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "backgammon-engine_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _wants_experimental(config):
    """True when the -m expression SELECTS tests because they carry gpu_experimental (`-m gpu_experimental`), not when it merely names the
    marker: `-m "not gpu_experimental"` / `-m "gpu and not gpu_experimental"` run the default build.  The expression is
    evaluated: it must hold for an item marked gpu_experimental only and fail for an unmarked one."""
    expr = config.getoption("-m") or ""
    if "gpu_experimental" not in expr:
        return False
    try:
        from _pytest.mark.expression import Expression
        e = Expression.compile(expr)
        return bool(e.evaluate(lambda name, **kw: name == "gpu_experimental")) and not bool(e.evaluate(lambda name, **kw: False))
    except Exception:                                    # a pytest without that module: fall back to asking for the marker alone
        return expr.strip() == "gpu_experimental"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_experimental: needs a real MI355X AND the -DBGAMD_EXPERIMENTAL build of the library (the kernels that "
                            "lost their A/B); run with -m gpu_experimental -- tools/round_check.sh does -- not part of -m gpu")


def pytest_collection_modifyitems(config, items):
    """gpu_experimental tests run only when asked for by name: `-m "not gpu"` (the CPU suite) must not pick them up."""
    if _wants_experimental(config):
        return
    skip = pytest.mark.skip(reason="experimental-build test: run with -m gpu_experimental")
    for it in items:
        if it.get_closest_marker("gpu_experimental"):
            it.add_marker(skip)


def pytest_sessionstart(session):
    """Always through build(): it is content-based (the library carries the digest of the sources it was compiled
    from) and a no-op when fresh, so the tests can never run a libbgamd.so that was not built from the tree they sit
    in (*.so is git-ignored but travels with gpurun snapshots).  hipcc cross-compiles gfx950 without a GPU."""
    import __graft_entry__
    if _wants_experimental(session.config):          # before anything imports the package (build() does): it binds BGAMD_LIB when it loads
        os.environ["BGAMD_LIB"] = __graft_entry__.build_experimental()
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
