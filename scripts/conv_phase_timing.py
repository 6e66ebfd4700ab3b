#!/usr/bin/env python3
"""Where a K-split conv launch spends its time: per-workgroup phase stamps (s_memrealtime, 10 ns ticks) written by a library
built with -DDT_CONV_TIMING.

    export DT_EXTRA_CFLAGS=-DDT_CONV_TIMING   # (also for the run: the library rebuilds itself when the flags change)
    python -m doubletake_amd._build --force && python scripts/conv_phase_timing.py
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

SHAPES = [(256, 256, 30, 40), (256, 256, 32, 32), (384, 384, 15, 20), (256, 256, 16, 32)]
PHASES = ["start skew", "setup (args, offsets)", "first loads + first K step", "remaining K steps", "LDS reduction",
          "epilogue (+ cross-WG)"]


def main():
    dev = gu.dev()
    stamps = torch.zeros(4096 * 24, dtype=torch.int64, device=dev)
    os.environ["DT_CONV_TIMING_PTR"] = hex(stamps.data_ptr())
    for cin, cout, h, w in SHAPES:
        conv = nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        x = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cin, h, w), 1)).to(dev))
        res = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cout, h, w), 2)).to(dev))
        for _ in range(6):
            stamps.zero_()
            ops.conv2d([(x, False)], conv, act=1, residual=res)
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy().reshape(-1, 24).astype(np.int64)
        raw = raw[raw[:, 0] != 0]
        t = raw[:, :6]
        t0 = t[:, 0].min()
        span = (t[:, 5].max() - t0) * 0.01
        d = np.concatenate([(t[:, :1] - t0), np.diff(t, axis=1)], axis=1) * 0.01  # us
        print(f"3x3 {cin}->{cout} {h}x{w}: {len(t)} workgroups, first start -> last end {span:.2f} us")
        for i, name in enumerate(PHASES):
            print(f"    {name:30s} mean {d[:, i].mean():6.2f}  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f} us")
        order = np.argsort(t[:, 0])
        late = t[order[-len(t) // 8:], 0].mean() - t0
        # per wave: end of the K loop relative to wave 0's, and the SIMD the wave runs on (HW_ID bits 4-5 on gfx9)
        wend = (raw[:, 8:16] - raw[:, 8:9]) * 0.01
        simd = (raw[:, 16:24] >> 4) & 3
        print("    K-loop end of waves 0-7 relative to wave 0 (mean us): " + " ".join(f"{v:5.2f}" for v in wend.mean(axis=0)))
        print("    slowest wave ends after wave 0 by: mean %.2f max %.2f us" % (wend.max(axis=1).mean(), wend.max()))
        counts = np.stack([(simd == k).sum(axis=1) for k in range(4)], axis=1)
        pats, freq = np.unique(np.sort(counts, axis=1), axis=0, return_counts=True)
        print("    waves per SIMD (sorted) patterns: " + ", ".join(f"{tuple(p_)} x{f}" for p_, f in zip(pats.tolist(), freq.tolist())))
        print(f"    latest eighth of the workgroups starts {late * 0.01:.2f} us after the first; ends: mean {(t[:, 5] - t0).mean() * 0.01:.2f} us")


if __name__ == "__main__":
    main()
