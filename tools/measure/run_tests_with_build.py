# Runs pytest against another build of the library (does a new test fail on the build it was written to catch?):
#   python tools/measure/run_tests_with_build.py old/liblinear_amd.so tests/test_gpu_parity.py -q -x -k n_run
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from linear_amd import api, build
api.SO = os.path.abspath(sys.argv[1])
build.needs_build = lambda: False
import pytest
sys.exit(pytest.main(sys.argv[2:]))
