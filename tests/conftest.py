import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (build container only)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session (single process, single card)."""
    import matchinglib_poselib_amd as mpa

    c = mpa.Context(0)
    yield c
    c.close()
