import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libpup_hip.so; fails loudly when it cannot be built."""
    from coolpuppy_amd import build, _ffi
    build.build_hip()
    return _ffi.lib()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import pileup_oracle
    pileup_oracle.build()
    return pileup_oracle
