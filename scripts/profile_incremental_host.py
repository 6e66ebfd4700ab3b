#!/usr/bin/env python3
"""cProfile of the host side of the incremental loop's FRAMES only (scripts/time_incremental.py's set-up excluded): where the
Python time of a frame goes.  DT_MODE = serial | programs (default)."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.abspath(os.path.dirname(__file__)))
os.environ.setdefault("DT_FRAMES", "168")
os.environ["DT_MODES"] = os.environ.get("DT_MODE", "programs")
import time_incremental as ti
from doubletake_amd import loops

pr = cProfile.Profile()
real = loops.run_incremental_scan
calls = [0]


def profiled(*a, **kw):
    calls[0] += 1
    if calls[0] < 2:      # the warm-up scan of a mode (records the programs)
        return real(*a, **kw)
    pr.enable()
    try:
        return real(*a, **kw)
    finally:
        pr.disable()


loops.run_incremental_scan = profiled
ti.main()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:56]))
