#!/usr/bin/env python
"""A/B helper: run a script of this repo against another build of libmhmr.so.
    python tools/run_with_lib.py build_ab/noslp bench.py --steps 10 ...
(the product always loads multi_hmr_amd/csrc/libmhmr.so; this only exists to compare builds on one GPU box)"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_hmr_amd import _lib
libdir, script = sys.argv[1], sys.argv[2]
if libdir not in ("", "default"):
    _lib.LIB_PATH = os.path.join(ROOT, libdir, "libmhmr.so")
    _lib.build = lambda *a, **k: _lib.LIB_PATH
sys.argv = [script] + sys.argv[3:]
runpy.run_path(os.path.join(ROOT, script), run_name="__main__")
