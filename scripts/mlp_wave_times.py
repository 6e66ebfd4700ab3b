#!/usr/bin/env python3
"""Per-wave start / end stamps of one launch of the fused MLP volume kernel at cfg2 (library built with -DDT_MLP_TIMING=2:
DOUBLETAKE_HIP_LIB=doubletake_amd/_lib/variants/timing.so).  Shows how evenly the 2048 resident waves finish."""
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
    t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 1))
    hm = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(hm.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
    gu.load_formula_mlp(hm.hint_mlp, [3, 12, 12, 1], 4)
    args, hd = gu.volume_call_args(t), gu.hint_dict(t)
    for _ in range(5):
        hm(**args, cv_depth_hint_dict=hd)
    torch.cuda.synchronize()
    n = 2048
    buf = (ctypes.c_ulonglong * (n * 4))()
    fn = _abi.lib().cdll.dt_debug_mlp_times
    fn.restype = ctypes.c_int
    assert fn(buf, n * 4) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t0 = a[:, 0].min()
    start, end, first = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0  # us (100 MHz counter)
    dur = end - start
    blk = a[:, 3] >> 8
    out = {"waves": n, "start_us_pct": np.percentile(start, [0, 50, 99, 100]).round(2).tolist(),
           "end_us_pct": np.percentile(end, [0, 1, 10, 50, 90, 99, 100]).round(2).tolist(),
           "dur_us_pct": np.percentile(dur, [0, 1, 10, 50, 90, 99, 100]).round(2).tolist(),
           "first_task_end_us_pct": np.percentile(first - start, [0, 50, 100]).round(2).tolist(),
           "mean_end_us": float(end.mean()), "max_end_us": float(end.max()),
           "idle_frac_after_mean_end": float(1 - end.mean() / end.max())}
    per_xcd = {int(x): round(float(end[(blk % 8) == x].mean()), 2) for x in range(8)}
    out["mean_end_us_per_xcd"] = per_xcd
    # waves of one SIMD pair (wave w and w+4 of a block share a SIMD): how far apart do partners finish?
    e = end.reshape(-1, 8)
    out["partner_end_gap_us_pct"] = np.percentile(np.abs(e[:, :4] - e[:, 4:]), [50, 90, 100]).round(2).tolist()
    # if the two waves of every SIMD shared their work dynamically, a SIMD would finish at about the mean of its pair; if a CU's
    # eight waves did, at the CU mean: how far apart are THOSE?  (what a pair- / CU-level split can and cannot recover)
    pair_mean = (e[:, :4] + e[:, 4:]) / 2.0
    cu_mean = e.mean(axis=1)
    out["simd_pair_mean_end_us_pct"] = np.percentile(pair_mean, [0, 10, 50, 90, 100]).round(1).tolist()
    out["cu_mean_end_us_pct"] = np.percentile(cu_mean, [0, 10, 50, 90, 100]).round(1).tolist()
    out["kernel_if_pairs_balanced_us"] = float(pair_mean.max())
    out["kernel_if_all_balanced_us"] = float(end.mean())
    print(json.dumps(out, indent=1))
    # progress curves: time per plane of the two waves of a SIMD while both run, and of the survivor alone
    pb = (ctypes.c_ulonglong * (n * 24))()
    fp = _abi.lib().cdll.dt_debug_mlp_progress
    fp.restype = ctypes.c_int
    assert fp(pb, n * 24) == 0
    pr = (np.frombuffer(pb, dtype=np.uint64).reshape(n, 24).astype(np.int64) - t0) / 100.0
    for blk_i in (3, 100):
        for wv in (0, 4):
            row = pr[blk_i * 8 + wv]
            row = row[row > 0][:20]
            print(f"block {blk_i} wave {wv}: plane-end times us", np.round(row, 1).tolist())
            print(f"   per-plane us", np.round(np.diff(row), 1).tolist())
    # all waves: mean time per plane while the partner is still running vs after it finished
    both, alone = [], []
    for i in range(n):
        blk_i, wv = divmod(i, 8)
        partner_end = end[blk_i * 8 + (wv + 4) % 8]
        row = pr[i]
        row = row[row > 0]
        for j in range(1, len(row)):
            (both if row[j] <= partner_end else alone).append(row[j] - row[j - 1])
    print("per-plane us while the SIMD partner runs: mean %.2f (n=%d); after it finished: mean %.2f (n=%d)" % (
        np.mean(both), len(both), np.mean(alone) if alone else float("nan"), len(alone)))


if __name__ == "__main__":
    main()
