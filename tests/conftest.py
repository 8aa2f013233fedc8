import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference; present only where oracle/_ref was built (or travelled)."""
    from pyoracle import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref/libslamref.so not present")
    return Ref()
