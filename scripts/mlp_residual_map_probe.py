#!/usr/bin/env python3
"""Where in the image are the volume kernel's waves slower / faster than the span plan predicts?  (-DDT_MLP_TIMING=1)
Per wave: actual duration / summed price, normalised per wave class (older / younger); averaged over an 6 x 8 grid of image
regions (by the position of the wave's middle unit) and over plane thirds.  Two frames: is the pattern a property of the image
position (stable) or of the frame?"""
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


def tile_order(h, w):
    """column strips of one tile width (32 pixels of a row), walked down / up alternately (csrc/cv_mlp_mfma.hip mlp_tile_order)"""
    tiles_per_row = (w + 31) // 32
    order = []
    for s in range(tiles_per_row):
        rows = range(h) if s % 2 == 0 else range(h - 1, -1, -1)
        for y in rows:
            order.append(y * tiles_per_row + s)
    return np.array(order)


def main():
    b, k, h, w, D = 1, 7, 120, 160, 64
    hm = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(hm.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
    gu.load_formula_mlp(hm.hint_mlp, [3, 12, 12, 1], 4)
    n = 2048
    L = _abi.lib().cdll
    L.dt_debug_mlp_times.restype = ctypes.c_int
    maps = {}
    for seed in (1, 1000):
        t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, seed))
        args, hd = gu.volume_call_args(t), gu.hint_dict(t)
        acc = None
        for rep in range(3):
            for _ in range(3):
                hm(**args, cv_depth_hint_dict=hd)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (n * 4))()
            assert L.dt_debug_mlp_times(buf, n * 4) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
            dur = (a[:, 1] - a[:, 0]) / 100.0
            acc = dur if acc is None else acc + dur
        dur = acc / 3
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        units = b * ((h * w + 31) // 32) * D
        n_ints = (cus * 8 + 2) // 2 * 2
        plan = hm._last_plan.cpu().numpy()
        bounds = plan[: 4 * n_ints].view(np.int32)[: n + 1].astype(np.int64)
        ngroups = (units + 255) // 256
        pref = plan[4 * (n_ints + (ngroups + 1) // 2 * 2):][: 4 * units].view(np.uint32).astype(np.int64)
        cost = pref - np.where(np.arange(units) % 256 == 0, 0, np.roll(pref, 1))
        csum = np.concatenate([[0], np.cumsum(cost)])
        pred = (csum[bounds[1:]] - csum[bounds[:-1]]).astype(float)
        u_mid = (bounds[1:] + bounds[:-1]) // 2
        # the kernel's block -> span mapping: logical block = (blockIdx % 8) * (nblk / 8) + blockIdx / 8; stamps are stored by wid of the LOGICAL block
        order = tile_order(h, w)
        tile = order[np.minimum(u_mid // D, len(order) - 1)]
        tiles_per_row = (w + 31) // 32
        ty, tx = tile // tiles_per_row, tile % tiles_per_row
        dthird = np.minimum((u_mid % D) * 3 // D, 2)
        wv = np.arange(n) % 8
        ratio = dur / pred
        for half in (0, 1):
            m = (wv // 4) == half
            ratio[m] /= ratio[m].mean()
        grid = np.full((6, tiles_per_row), np.nan)
        for gy in range(6):
            for gx in range(tiles_per_row):
                m = (ty * 6 // h == gy) & (tx == gx)
                if m.sum() >= 4:
                    grid[gy, gx] = ratio[m].mean()
        maps[seed] = grid
        print(f"seed {seed}: residual (actual / predicted, 1.00 = average) by image region (rows: 6 bands top to bottom; columns: {tiles_per_row} tile columns)")
        for row in grid:
            print("   " + " ".join("  .  " if np.isnan(v) else f"{v:5.3f}" for v in row))
        print("   by plane third (near, middle, far):", [round(float(ratio[dthird == i].mean()), 4) for i in range(3)],
              " rms", round(float(ratio.std()), 4))
    a, c = maps[1].ravel(), maps[1000].ravel()
    ok = ~np.isnan(a) & ~np.isnan(c)
    print("correlation of the two frames' region maps:", round(float(np.corrcoef(a[ok], c[ok])[0, 1]), 3))


if __name__ == "__main__":
    main()
