#!/usr/bin/env python3
"""GPU box: bench.py with a variant build of the library (ASDR_TOOLS_LIB=<path>; one library per process -- tools/_variant.py).
    ASDR_TOOLS_LIB=audiosdr_amd/variants/libasdr_x.so python3 tools/bench_variant.py [bench.py flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402,F401
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
