#!/usr/bin/env python3
"""Runs bench.py against an alternative build of the library (tuning experiments): tools/bench_variant.py <lib.so> [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linear_amd import api
api.SO = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
