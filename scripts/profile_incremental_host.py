#!/usr/bin/env python3
"""cProfile of the host side of the incremental loop (scripts/time_incremental.py, serial mode): where the Python time of a
frame goes.  The loop is host-bound when host issue time per frame exceeds the GPU time."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.abspath(os.path.dirname(__file__)))
os.environ.setdefault("DT_FRAMES", "120")
import time_incremental as ti

pr = cProfile.Profile()
pr.enable()
ti.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
