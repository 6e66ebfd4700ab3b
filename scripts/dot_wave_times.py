#!/usr/bin/env python3
"""Per-wave start / end stamps of one launch of the dot-product volume kernel at cfg2 (library variant built with
-DDT_DOT_TIMING): duration of a wave by plane group -- are the near-plane groups the kernel's critical path?"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
import torch

import gpu_util as gu
from doubletake_amd import _abi
from doubletake_amd.modules.cost_volume import CostVolumeManager
from doubletake_amd.utils import synthetic as syn


def main():
    b, k, h, w, D = 1, 7, 120, 160, 64
    t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 1))
    m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
    args = gu.volume_call_args(t)
    for _ in range(5):
        m(**args)
    torch.cuda.synchronize()
    n = 16384
    buf = (ctypes.c_ulonglong * (n * 3))()
    fn = _abi.lib().cdll.dt_debug_dot_times
    fn.restype = ctypes.c_int
    assert fn(buf, n * 3) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    grp = a[:, 2] >> 32
    print("waves", len(a), "kernel span us", round(float(end.max()), 2))
    for g in sorted(set(grp.tolist())):
        sel = grp == g
        d = end[sel] - start[sel]
        print(f"plane group {g:2d}: waves {sel.sum():4d}  start {start[sel].min():6.1f}..{start[sel].max():6.1f}  "
              f"duration mean {d.mean():6.2f} max {d.max():6.2f}  last end {end[sel].max():6.1f}")


if __name__ == "__main__":
    main()
