#!/usr/bin/env python3
"""Span-level check of the plan (library built with -DDT_MLP_TIMING=1): actual duration of every wave against the summed prices of
its span; what do the residuals correlate with?"""
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
from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager
from doubletake_amd.utils import synthetic as syn


def main():
    b, k, h, w, D = 1, 7, 120, 160, 64
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    hm = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(hm.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
    gu.load_formula_mlp(hm.hint_mlp, [3, 12, 12, 1], 4)
    t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, seed))
    args, hd = gu.volume_call_args(t), gu.hint_dict(t)
    for _ in range(4):
        hm(**args, cv_depth_hint_dict=hd)
    torch.cuda.synchronize()
    n = 2048
    L = _abi.lib().cdll
    buf = (ctypes.c_ulonglong * (n * 4))()
    L.dt_debug_mlp_times.restype = ctypes.c_int
    assert L.dt_debug_mlp_times(buf, n * 4) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t0 = a[:, 0].min()
    start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    dur = end - start
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    units = b * ((h * w + 31) // 32) * D
    n_ints = (cus * 8 + 2) // 2 * 2
    plan = hm._last_plan.cpu().numpy()
    bounds = plan[: 4 * n_ints].view(np.int32)[: n + 1].astype(np.int64)
    ngroups = (units + 255) // 256
    pref = plan[4 * (n_ints + (ngroups + 1) // 2 * 2):][: 4 * units].view(np.uint32).astype(np.int64)
    cost = pref - np.where(np.arange(units) % 256 == 0, 0, np.roll(pref, 1))
    csum = np.concatenate([[0], np.cumsum(cost)])
    pred = csum[bounds[1:]] - csum[bounds[:-1]]
    length = bounds[1:] - bounds[:-1]
    u_mid = (bounds[1:] + bounds[:-1]) // 2
    otile = u_mid // D          # position in the tile ORDER (not the image position)
    d_mid = (u_mid % D) / D
    wv = np.arange(n) % 8
    out = {"seed": seed, "kernel_us": round(float(end.max()), 1), "mean_end_us": round(float(end.mean()), 1),
           "span_len_pct": np.percentile(length, [0, 50, 100]).tolist()}
    for half, name in ((0, "older"), (1, "younger")):
        m = (wv // 4) == half
        ratio = dur[m] / pred[m]
        ratio = ratio / ratio.mean()
        feats = {"d_mid": d_mid[m], "tile_order_pos": otile[m] / otile.max(), "span_len": length[m].astype(float),
                 "block": (np.arange(n)[m] // 8) / 256.0, "start_us": start[m]}
        out[name] = {"ratio_std": round(float(ratio.std()), 4), "ratio_pct_1_50_99": np.percentile(ratio, [1, 50, 99]).round(3).tolist(),
                     "corr": {k: round(float(np.corrcoef(v, ratio)[0, 1]), 3) for k, v in feats.items()}}
    # pair level: (older + younger) mean end vs launch mean
    e = end.reshape(-1, 8)
    pair = (e[:, :4] + e[:, 4:]) / 2.0
    p_pred = pred.reshape(-1, 8)
    out["pair_mean_end_std_us"] = round(float(pair.std()), 2)
    out["pair_pred_sum_rel_std"] = round(float((p_pred[:, :4] + p_pred[:, 4:]).std() / (p_pred[:, :4] + p_pred[:, 4:]).mean()), 4)
    out["older_end_minus_younger_end_us_pct"] = np.percentile(e[:, :4] - e[:, 4:], [1, 10, 50, 90, 99]).round(1).tolist()
    out["end_us_pct_older"] = np.percentile(e[:, :4], [0, 10, 50, 90, 100]).round(1).tolist()
    out["end_us_pct_younger"] = np.percentile(e[:, 4:], [0, 10, 50, 90, 100]).round(1).tolist()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
