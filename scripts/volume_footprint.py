#!/usr/bin/env python3
"""Source-feature footprint of the fused volume kernel per XCD (CPU, numpy; no GPU, no oracle).

Every XCD has a private L2, and the kernel gives each XCD one contiguous eighth of the tile order.  The texels (64 B: one
NHWC pixel of 16 floats) an eighth's epipolar segments touch have to reach that XCD's L2 at least once, so the sum over the
eight XCDs of the unique texels touched -- not one copy of the source maps -- is the floor of the kernel's source fetch traffic
(what FETCH_SIZE counts at the L2 -> fabric boundary).  This script evaluates that sum on bench.py's geometry (seed 1000) for
the row-major tile order of rounds 1-3 and for the column-strip orders of `mlp_tile_order` (cv_mlp_mfma.hip), which is how the
strip order was chosen.  Projection as in cv_geometry.hpp (P = K_src @ src_T_cur; log-spaced planes), bilinear 2x2 taps."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

from doubletake_amd.utils import synthetic as syn

h, w, K, D = 120, 160, 7, 64
inp = syn.volume_inputs(1, K, h, w, 16, 1000)
ext, Ks, invK = inp["src_extrinsics"][0], inp["src_Ks"][0], inp["cur_invK"][0]
mn, mx = float(np.ravel(inp["min_depth"])[0]), float(np.ravel(inp["max_depth"])[0])
planes = np.exp(np.log(mn) + np.log(mx / mn) * np.linspace(0, 1, D))
ys, xs = np.mgrid[0:h, 0:w]
rays = invK[:3, :3] @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)], 0).astype(np.float64)
NT = h * w // 32
TY, TX = (np.arange(NT) * 32) // w, ((np.arange(NT) * 32) % w) // 32


def footprint_mb(xcd_of_pixel, nx=8):
    total = 0
    for k in range(K):
        P = (Ks[k] @ ext[k])[:3]
        hit = np.zeros((nx, h, w), bool)
        for d in planes:
            q = P @ np.concatenate([rays * d, np.ones((1, h * w))], 0)
            ok = q[2] > 1e-6
            z = np.where(ok, q[2], 1.0)
            x0, y0 = np.floor(q[0] / z).astype(int), np.floor(q[1] / z).astype(int)
            for dx in (0, 1):
                for dy in (0, 1):
                    xx, yy = x0 + dx, y0 + dy
                    m = ok & (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
                    hit[xcd_of_pixel[m], yy[m], xx[m]] = True
        total += int(hit.sum())
    return total * 64 / 1e6


def strips(num_row_blocks, boustrophedon):
    hh = -(-h // num_row_blocks)
    blk = TY // hh
    col = np.where(blk % 2 == 1, TX.max() - TX, TX) if boustrophedon else TX
    row = np.where(col % 2 == 1, -TY, TY) if boustrophedon else TY
    order = np.lexsort((row, col, blk))
    rank = np.empty(NT, int)
    rank[order] = np.arange(NT)
    return np.repeat(rank * 8 // NT, 32)[: h * w]


pid = np.arange(h * w)
print(f"one copy of what the frame touches          {footprint_mb(np.zeros(h * w, int), 1):6.2f} MB")
print(f"row-major order (an XCD = 15 rows x 160)    {footprint_mb(pid // (h * w // 8)):6.2f} MB")
for nb in (1, 2, 4):
    for bs in (False, True):
        print(f"column strips, {nb} row block(s), boustrophedon={bs!s:5}  {footprint_mb(strips(nb, bs)):6.2f} MB")
