import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def lc(sub=None):
    """import landing-controller_amd[.sub] (the package directory name carries a hyphen)."""
    name = "landing-controller_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
