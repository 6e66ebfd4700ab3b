#!/usr/bin/env python3
"""Where a Winograd conv launch spends its time, per K step: phase sums of one wave per workgroup (s_memrealtime, 10 ns ticks)
written by a library variant built with -DDT_CONV_TIMING (scripts/build_variant.py timing "-DDT_CONV_TIMING").

    DOUBLETAKE_HIP_LIB=doubletake_amd/_lib/variants/timing.so python scripts/wino_phase_timing.py
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
import torch
import torch.nn as nn

import gpu_util as gu
from doubletake_amd.modules import conv_ops as ops
from doubletake_amd.utils import synthetic as syn

SHAPES = [(64, 64, 240, 320), (128, 64, 240, 320), (64, 64, 120, 160), (128, 128, 60, 80)]
PHASES = ["wait prefetched global data + patch->LDS", "barrier", "issue next global loads", "LDS window reads + transform", "16 MFMAs issue"]


def main():
    dev = gu.dev()
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    os.environ["DT_CONV_TIMING_PTR"] = hex(stamps.data_ptr())
    for cin, cout, h, w in SHAPES:
        conv = nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        x = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cin, h, w), 1)).to(dev))
        for _ in range(4):
            stamps.zero_()
            ops.conv2d([(x, False)], conv, act=1, impl="wino")
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy().reshape(-1, 8).astype(np.int64)
        raw = raw[raw[:, 0] != 0]
        iters = cin // 8
        t0 = raw[:, 0].min()
        span = (raw[:, 6].max() - t0) * 0.01
        life = (raw[:, 6] - raw[:, 0]) * 0.01
        print(f"3x3 {cin}->{cout} {h}x{w}: {len(raw)} workgroups, first start -> last end {span:.2f} us; K loop of one workgroup "
              f"(incl. prologue) mean {life.mean():.2f} us; {iters} K steps; MFMA-only time of a K step = 0.43 us at 2.4 GHz")
        tot = raw[:, 1:6].sum(axis=1).mean() * 0.01
        for i, name in enumerate(PHASES):
            v = raw[:, 1 + i] * 0.01
            print(f"    {name:44s} {v.mean():6.2f} us per workgroup = {v.mean() / iters * 1e3:6.0f} ns per K step ({100 * v.mean() / tot:4.1f} %)")


if __name__ == "__main__":
    main()
