#!/usr/bin/env python3
"""Per-layer timing of the conv primitive (hipEvents over back-to-back launches)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch
import torch.nn as nn

import gpu_util as gu
from doubletake_amd.modules import conv_ops as ops
from doubletake_amd.utils import synthetic as syn

LAYERS_ALL = [
    # name, cin, cout, k, stride, h, w
    ("L0 3x3 64->64 120x160", 64, 64, 3, 1, 120, 160),
    ("L0 3x3 128->64 120x160", 128, 64, 3, 1, 120, 160),
    ("L1 3x3 128->128 60x80", 128, 128, 3, 1, 60, 80),
    ("L2 3x3 256->256 30x40", 256, 256, 3, 1, 30, 40),
    ("L3 3x3 384->384 15x20", 384, 384, 3, 1, 15, 20),
    ("L3 3x3 896->384 15x20", 896, 384, 3, 1, 15, 20),
    ("s2 3x3 64->128 120x160", 64, 128, 3, 2, 120, 160),
    ("S0 3x3 128->64 240x320", 128, 64, 3, 1, 240, 320),
    ("S0 3x3 64->64 240x320", 64, 64, 3, 1, 240, 320),
    ("S0 1x1 64->128 240x320", 64, 128, 1, 1, 240, 320),
    ("S0 1x1 128->128 240x320", 128, 128, 1, 1, 240, 320),
    ("L3 1x1 896->384 15x20", 896, 384, 1, 1, 15, 20),
]


if os.environ.get("DT_LAYER_SWEEP"):
    # fixed cost vs per-K cost of one launch: same output, growing input channels
    LAYERS_ALL = ([(f"sweep 3x3 {c}->64 120x160", c, 64, 3, 1, 120, 160) for c in (8, 16, 32, 64, 128, 256)] +
                  [(f"sweep 3x3 {c}->128 60x80", c, 128, 3, 1, 60, 80) for c in (8, 32, 128, 256)] +
                  [(f"sweep 3x3 {c}->256 30x40", c, 256, 3, 1, 30, 40) for c in (16, 64, 256, 512)] +
                  [(f"sweep 3x3 {c}->384 15x20", c, 384, 3, 1, 15, 20) for c in (16, 128, 384, 896)])
LAYERS = [l for l in LAYERS_ALL if not os.environ.get('DT_LAYER_FILTER') or any(f in l[0] for f in os.environ['DT_LAYER_FILTER'].split(','))]


def main():
    dev = gu.dev()
    res = {}
    for name, cin, cout, k, st, h, w in LAYERS:
        conv = nn.Conv2d(cin, cout, k, stride=st, padding=k // 2).to(dev)
        x = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cin, h, w), 1)).to(dev))
        for _ in range(5):
            ops.conv2d([(x, False)], conv, act=1)
        torch.cuda.synchronize()
        n = 40
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            ops.conv2d([(x, False)], conv, act=1)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        ho, wo = (h + st - 1) // st, (w + st - 1) // st
        gf = 2.0 * ho * wo * cout * cin * k * k / 1e9
        res[name] = dict(us=round(us, 1), gflop=round(gf, 3), tflops=round(gf / us * 1e-3 * 1e3, 1))
        print(f"{name:28s} {us:8.1f} us  {gf:7.3f} GF  {gf / us * 1e3:6.1f} TF/s")
    json.dump(res, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "conv_layers.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
