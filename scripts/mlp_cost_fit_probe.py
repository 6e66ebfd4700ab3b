#!/usr/bin/env python3
"""How well does the span plan's unit price predict the time a wave really spends on a (tile, plane) unit?
(library built with -DDT_MLP_TIMING=2).  Per-unit durations from the kernel's progress stamps (first 23 units of every wave), the
plan's prices from its scratch; least-squares fit of duration against views seen, plane index and position in the tile, for the
older and the younger wave of a SIMD pair separately (only units finished while the partner was still running)."""
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
    pb = (ctypes.c_ulonglong * (n * 24))()
    L.dt_debug_mlp_progress.restype = ctypes.c_int
    assert L.dt_debug_mlp_progress(pb, n * 24) == 0
    pr = (np.frombuffer(pb, dtype=np.uint64).reshape(n, 24).astype(np.int64) - t0) / 100.0
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    units = b * ((h * w + 31) // 32) * D
    n_ints = (cus * 8 + 2) // 2 * 2
    plan = hm._last_plan.cpu().numpy()
    bounds = plan[: 4 * n_ints].view(np.int32)[: n + 1].astype(np.int64)
    ngroups = (units + 255) // 256
    pref = plan[4 * (n_ints + (ngroups + 1) // 2 * 2):][: 4 * units].view(np.uint32).astype(np.int64)
    cost = pref - np.where(np.arange(units) % 256 == 0, 0, np.roll(pref, 1))
    seen = (cost - (290 + 27 * k)) // 32
    rows = {0: [], 1: []}
    for i in range(n):
        blk, wv = divmod(i, 8)
        partner_end = end[blk * 8 + (wv + 4) % 8]
        row = pr[i]
        row = row[row > 0]
        u0 = bounds[i]
        for j in range(1, len(row)):
            u = u0 + j
            if u >= bounds[i + 1] or row[j] > partner_end:
                break
            d = u % D
            rows[wv // 4].append((row[j] - row[j - 1], seen[u], d, 1.0 if d == 0 else 0.0))
    out = {"seed": seed, "kernel_us": round(float(end.max()), 1), "mean_end_us": round(float(end.mean()), 1)}
    for half, name in ((0, "older"), (1, "younger")):
        r = np.array(rows[half])
        y = r[:, 0]
        X = np.stack([np.ones(len(r)), r[:, 1], r[:, 2] / D, r[:, 3]], 1)
        coef, *_ = np.linalg.lstsq(X, y, rcond=None)
        res = y - X @ coef
        X1 = X[:, :2]
        c1, *_ = np.linalg.lstsq(X1, y, rcond=None)
        out[name] = {"n": len(r), "mean_us": round(float(y.mean()), 2), "std_us": round(float(y.std()), 2),
                     "fit_const_seen_plane_newtile": np.round(coef, 3).tolist(), "residual_std_us": round(float(res.std()), 2),
                     "fit_const_seen_only": np.round(c1, 3).tolist(), "residual_std_seen_only": round(float((y - X1 @ c1).std()), 2),
                     "mean_by_seen": {int(s): round(float(y[r[:, 1] == s].mean()), 2) for s in np.unique(r[:, 1])},
                     "plan_ratio_all_vs_none": round((290 + 27 * k + 32 * k) / (290 + 27 * k), 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
