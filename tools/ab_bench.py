#!/usr/bin/env python3
"""Diagnostic: bench.py against a variant library (--lib path), for A/B runs on one box in one call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagesegmentation_amd import _lib  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
import bench  # noqa: E402

bench.main()
