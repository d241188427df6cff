import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (liblsx.so and the C oracle are git-ignored): build them
    once, the same way the driver's build check does (hipcc cross-compiles gfx950 without a GPU)."""
    lib = os.path.join(ROOT, "linalg_solver_amd", "liblsx.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
