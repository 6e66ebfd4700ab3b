#!/usr/bin/env python3
"""hipEvent timing of ResnetMatchingEncoder on the 1+7 images of one 640x480 keyframe tuple."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch

import gpu_util as gu
from doubletake_amd.modules.networks import ResnetMatchingEncoder
from doubletake_amd.utils import synthetic as syn


def main():
    dev = gu.dev()
    out = {}
    only = os.environ.get("DT_ENC_ONLY")  # e.g. "1,8": antialiased=True, n=8 only (for a clean kernel trace)
    for aa in (True, False):
        m = ResnetMatchingEncoder(18, 16, pretrained=False, antialiased=aa).to(dev)
        for n in (1, 8):
            if only and (str(int(aa)), str(n)) != tuple(only.split(",")):
                continue
            img = torch.from_numpy(syn.hash_normalish((n, 3, 480, 640), 3)).to(dev)
            for _ in range(5):
                m(img)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                m(img)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            gf = n * (2 * 76800 * 64 * 147 + 4 * 2 * 19200 * 64 * 64 * 9 + 2 * 19200 * 64 * 128 + 2 * 19200 * 128 * 16 * 9) / 1e9
            out[f"antialiased={aa} n={n}"] = dict(ms=round(ms, 3), gflop=round(gf, 2), tflops=round(gf / ms, 1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
