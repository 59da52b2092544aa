import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name is not an identifier -> importlib)."""
    import __graft_entry__ as ge
    return ge.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def engine(pkg):
    return pkg.ctc_segmentation.default_engine(0)
