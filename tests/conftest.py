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


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first(request):
    """GPU runs: torch initialises its HIP context before the first test creates a library context (a test that hands device blobs to torch
    must not depend on an earlier test having done so)."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
