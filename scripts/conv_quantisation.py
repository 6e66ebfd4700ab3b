#!/usr/bin/env python3
"""Workgroup-count quantisation of the conv kernels: the same 3x3 layer on maps whose block count sweeps across multiples of
the 256 CUs.  Run under `rocprofv3 --kernel-trace --output-format csv` and summarise with `--summarise <kernel_trace.csv>`
(GPU-side durations; host pacing does not matter)."""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))

SHAPES = [  # cin, cout, h, w
    (256, 256, 16, 32), (256, 256, 32, 32), (256, 256, 30, 40), (256, 256, 32, 48), (256, 256, 32, 64), (256, 256, 48, 64),
    (384, 384, 15, 20), (384, 384, 16, 24), (384, 384, 16, 32), (384, 384, 32, 32),
    (128, 128, 60, 80), (128, 128, 64, 64), (64, 64, 120, 160), (64, 64, 128, 128),
]


def summarise(path):
    rows = list(csv.DictReader(open(path)))
    by = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv" not in name or "pack" in name:
            continue
        wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"])
        grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
        by[(name, grid // wg, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, wgs, wg), d in sorted(by.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        d.sort()
        print(f"{name[:60]:60s} wgs={wgs:5d} x{wg:4d}  n={len(d):3d}  median {d[len(d) // 2]:7.2f} us  min {d[0]:7.2f}")


def main():
    import torch
    import torch.nn as nn

    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.utils import synthetic as syn

    dev = gu.dev()
    for cin, cout, h, w in SHAPES:
        conv = nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        x = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cin, h, w), 1)).to(dev))
        for _ in range(12):
            ops.conv2d([(x, False)], conv, act=1)
        torch.cuda.synchronize()
        print(f"{cin}->{cout} {h}x{w}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        main()
