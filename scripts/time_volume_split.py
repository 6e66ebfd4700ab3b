#!/usr/bin/env python3
"""Exact-fp32 vs opt-in split-precision (fp16 hi/lo) MLP volume kernel at cfg2 (640x480, K=7, D=64, B=1) and cfg3 (B=8)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
import torch

import gpu_util as gu
from doubletake_amd.modules import cost_volume as cvmod
from doubletake_amd.utils import synthetic as syn


def main():
    out = {}
    cfgs = {"cfg2": (1, 7, 120, 160, 64), "cfg3_b8": (8, 7, 96, 128, 64)}
    if "--once" in sys.argv:
        cfgs = {"cfg2": cfgs["cfg2"]}
    for name, (b, k, h, w, D) in cfgs.items():
        t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 1000))
        m = cvmod.FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
        gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 31)
        gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 32)
        res = {}
        vols = {}
        for prec in ("fp32", "split16"):
            m.precision = prec
            evs = []

            def hook(tag):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)

            call = lambda: m(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t))
            for _ in range(3):
                vols[prec] = call()[0]
            cvmod.FeatureVolumeManager._event_hook = staticmethod(hook)
            for _ in range(2 if "--once" in sys.argv else 20):
                call()
            torch.cuda.synchronize()
            cvmod.FeatureVolumeManager._event_hook = None
            res[prec + "_ms"] = float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(0, len(evs), 2)]))
        d = (vols["fp32"] - vols["split16"]).abs()
        res["max_abs_diff"], res["mean_abs_diff"] = d.max().item(), d.mean().item()
        res["speedup"] = res["fp32_ms"] / res["split16_ms"]
        out[name] = res
        print(name, res, flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "time_volume_split.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
