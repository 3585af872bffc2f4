import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle

    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def emu():
    import emu_ffi

    emu_ffi.load()
    return emu_ffi


@pytest.fixture(scope="session")
def ctx():
    """HIP context on device 0.  Fails loudly (no skip, no CPU fallback) when the extension or GPU is missing."""
    from limo_amd import ba

    return ba.Context(0)
