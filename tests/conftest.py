import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build what is missing (fresh checkout): the HIP library cross-compiles without a GPU, the C++
    host mirror and the oracle are plain g++/gcc.  Up-to-date artefacts are left alone."""
    from gfasort_amd import build as B
    from oracle import oracle as O
    B.build_hip()
    B.build_host()
    O.build()


@pytest.fixture(scope="session")
def data_dir():
    return os.path.join(ROOT, "tests", "data")
