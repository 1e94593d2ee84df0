#!/usr/bin/env python3
"""Run a probe script against another build of the library (A/B of compile-time choices inside ONE gpurun call, so
that box-to-box spread does not blur the comparison):  python scripts/ab_lib.py <libfq_hip variant .so> <script> [args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-quantity_amd", "quantity"))
from common.quantity import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
