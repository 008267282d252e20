import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds (if needed) and returns the oracle binding module. TEST-ONLY checker."""
    from oracle import pyorc
    pyorc.build(ref=os.path.exists("/root/reference/src/pmpfinder.cpp"))
    return pyorc


_case_cache = {}


@pytest.fixture(scope="session")
def case_inputs():
    from tests import cases

    def get(name):
        if name not in _case_cache:
            _case_cache[name] = (cases.CASES[name][0] if name in cases.CASES else cases.CASES_I2[name][0] if name in cases.CASES_I2 else cases.CASES_G50[name][0])()
        return _case_cache[name]
    return get
