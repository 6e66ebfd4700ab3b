#!/usr/bin/env python3
"""Replica vs voxel-slab fusion of the final volume (0.02 m over 8 x 8 x 3.2 m = 400 x 400 x 160, and 0.04 m): HIP-event
time of ONE multi-frame integrate call of N gathered frames into the whole volume (what every rank does per step in replica
mode) and into an X/8 slab (slab mode on 8 GPUs), plus the byte size of the end-of-pass slab gather.  DESIGN.md section 6."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from doubletake_amd import parallel as par
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn


def main():
    dev = torch.device("cuda:0")
    room = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    H2, W2 = 240, 320
    depth, K, T = syn.tsdf_frames(16, H2, W2, seed=5, bounds=room)
    d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
    out = {}
    for res in (0.04, 0.02):
        f = OurFuser(None, res, 3.0, bounds=room)
        X = int(f.tsdf_fuser_pred.tsdf.tsdf_values.shape[0])
        for label, rng in (("whole", None), ("slab_1_of_8", par.slab_bounds(X, 8, 3))):
            f.tsdf_fuser_pred.x_range = rng
            for n in (1, 2, 4, 8, 16):
                for _ in range(3):
                    f.fuse_frames(d[:n], k[:n], t[:n], None)
                torch.cuda.synchronize()
                evs = []
                for _ in range(10):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    f.fuse_frames(d[:n], k[:n], t[:n], None)
                    b.record()
                    evs.append((a, b))
                torch.cuda.synchronize()
                out[f"{res:.2f}m_{label}_{n}_frames_ms"] = round(float(np.median([a.elapsed_time(b) for a, b in evs])), 4)
        vol = f.tsdf_fuser_pred.tsdf
        out[f"{res:.2f}m_gather_bytes_per_rank_of_8"] = int((vol.tsdf_values.numel() * 2 * 2 + vol.voxel_bitmap.numel() * 4))
        del f
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
