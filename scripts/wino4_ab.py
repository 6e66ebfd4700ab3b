#!/usr/bin/env python3
"""Per-layer A/B of the Winograd kernels: F(2x2,3x3) (conv_wino_kernel) against F(4x4,3x3) (conv_wino4_kernel) on the 3x3 stride-1
layer shapes of the conv stacks, batch 1 and 8.  HIP events around 100 back-to-back launches of each (the GPU stays busy: the
figure is the kernel's, not the host's), error of both against the direct general-shape kernel."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from doubletake_amd.modules import conv_ops as ops
from doubletake_amd.utils import synthetic as syn

dev = torch.device("cuda:0")
SHAPES = [  # (n, [source channels], cout, h, w)
    (1, [64], 64, 240, 320), (1, [64, 64], 64, 240, 320), (1, [128], 64, 240, 320),
    (1, [64], 64, 120, 160), (1, [128, 64], 128, 120, 160), (1, [128], 128, 60, 80), (1, [256], 256, 30, 40),
    (8, [64], 64, 192, 256), (8, [128], 64, 192, 256), (8, [64], 64, 96, 128), (8, [128], 128, 48, 64), (8, [256], 256, 24, 32),
    (8, [64, 64, 64], 64, 192, 256), (2, [64], 64, 256, 192),
]


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


if os.environ.get("DT_W4_SHAPES") == "short":
    SHAPES = [(1, [64], 64, 240, 320), (8, [128], 64, 192, 256), (8, [128], 128, 48, 64)]
print(f"{'shape':44s} {'blocks F4':>9s} {'F2 us':>8s} {'F4 us':>8s} {'F2/F4':>6s} {'direct-eq TF F2':>15s} {'F4':>6s} {'err F2':>9s} {'err F4':>9s}")
for n, cs, cout, h, w in SHAPES:
    cin = sum(cs)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(syn.hash_normalish(tuple(conv.weight.shape), 5) * (1.0 / np.sqrt(9.0 * cin))))
    srcs = [(ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, c, h, w), 1 + i)).to(dev)), False) for i, c in enumerate(cs)]
    want = ops.conv2d(srcs, conv, act=ops.ACT_ELU, impl="simple")
    res = {}
    for tag, thr in (("F2", 0), ("F4", 1)):
        ops.WINO4_MIN_BLOCKS = thr
        got = ops.conv2d(srcs, conv, act=ops.ACT_ELU, impl="wino")
        res[tag] = (timed(lambda: ops.conv2d(srcs, conv, act=ops.ACT_ELU, impl="wino")), float((got - want).abs().max()))
    flops = 2.0 * n * h * w * cout * cin * 9
    blocks = n * ((h + 15) // 16) * ((w + 15) // 16) * (cout // 32)
    print(f"{str((n, cs, cout, h, w)):44s} {blocks:9d} {res['F2'][0]:8.1f} {res['F4'][0]:8.1f} {res['F2'][0] / res['F4'][0]:6.2f} "
          f"{flops / res['F2'][0] / 1e6:15.1f} {flops / res['F4'][0] / 1e6:6.1f} {res['F2'][1]:9.2e} {res['F4'][1]:9.2e}")
