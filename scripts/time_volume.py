#!/usr/bin/env python3
"""Quick hipEvent timing of the volume kernels at BASELINE configs[1] (640x480, K7, D64)."""
import json
import sys
import os

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
import torch

import gpu_util as gu
from doubletake_amd.modules.cost_volume import CostVolumeManager, FeatureMeshHintVolumeManager
from doubletake_amd.utils import synthetic as syn


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    out = {}
    for name, (b, k, h, w, D) in {"cfg1": (1, 2, 64, 80, 32), "cfg2": (1, 7, 120, 160, 64), "cfg3": (8, 7, 96, 128, 64)}.items():
        t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 1))
        m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
        args = gu.volume_call_args(t)
        out[f"{name}_dot_ms"] = timeit(lambda: m(**args))
        hm = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
        gu.load_formula_mlp(hm.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
        gu.load_formula_mlp(hm.hint_mlp, [3, 12, 12, 1], 4)
        hd = gu.hint_dict(t)
        out[f"{name}_hint_mfma_ms"] = timeit(lambda: hm(**args, cv_depth_hint_dict=hd))
        pairs = b * D * h * w
        flops = 2.0 * pairs * ((20 * (k + 1) + 6 * k) * 128 + 128 * 128 + 128 + 192)
        out[f"{name}_hint_tflops_algo"] = flops / (out[f"{name}_hint_mfma_ms"][0] * 1e-3) / 1e12
        if name != "cfg3":
            out[f"{name}_hint_simple_ms"] = timeit(
                lambda: hm._forward_impl(**args, cv_depth_hint_dict=hd, depth_planes_bdhw=None, return_mask=False,
                                         _impl="simple"), n=3, warm=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
