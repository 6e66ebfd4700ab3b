#!/usr/bin/env python3
"""Soak test of the cross-workgroup K reduction (csrc/conv.hip: write-through partial blocks + arrival counter, no fences):
thousands of launches of the layers that use it, from two streams at once and interleaved with a bandwidth-heavy kernel, every
output compared bit for bit with the first one.  A stale or torn partial read would show up as a mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch
import torch.nn as nn

import gpu_util as gu
from doubletake_amd.modules import conv_ops as ops
from doubletake_amd.utils import synthetic as syn

CASES = [  # cin, cout, h, w: direct K-split x2 (transposed), tail split, Winograd x2, Winograd tail split
    (384, 384, 15, 20), (256, 256, 16, 32), (896, 384, 15, 20), (256, 256, 30, 40), (640, 256, 30, 40), (64, 64, 120, 160),
]


def run(iters, verbose=True):
    """Returns the number of launches whose output differed from the first run of the same layer."""
    dev = gu.dev()
    layers = []
    for i, (cin, cout, h, w) in enumerate(CASES):
        conv = nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        x = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cin, h, w), 10 + i)).to(dev))
        res = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((1, cout, h, w), 20 + i)).to(dev))
        want = ops.conv2d([(x, False)], conv, act=1, residual=res).clone()
        simple = ops.conv2d([(x, False)], conv, act=1, residual=res, impl="simple")
        err = (want - simple).abs().max().item()
        assert err < 4e-6 * max(simple.abs().max().item(), 5.0), err
        layers.append((conv, x, res, want))
    noise = torch.empty(64 << 20, device=dev)  # 256 MB: evicts the L2s between launches
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    bad = 0
    t0 = time.time()
    pending = []
    for it in range(iters):
        conv, x, res, want = layers[it % len(layers)]
        with torch.cuda.stream(streams[it % 2]):
            if it % 7 == 0:
                noise.add_(1.0)
            pending.append((ops.conv2d([(x, False)], conv, act=1, residual=res), want, it))
        if len(pending) >= 64:
            torch.cuda.synchronize()
            for got, w_, i_ in pending:
                if not torch.equal(got, w_):
                    bad += 1
                    print(f"MISMATCH at launch {i_}: max diff {(got - w_).abs().max().item():.3e}")
            pending = []
    torch.cuda.synchronize()
    for got, w_, i_ in pending:
        bad += 0 if torch.equal(got, w_) else 1
    if verbose:
        print(f"{iters} launches on 2 streams, {len(CASES)} layer shapes: {bad} mismatches ({time.time() - t0:.1f} s)")
    return bad


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    sys.exit(1 if run(iters) else 0)


if __name__ == "__main__":
    main()
