#!/usr/bin/env python3
"""A few 16-frame integrate calls into the 0.02 m volume of the 8 x 8 x 3.2 m room (for rocprofv3 --kernel-trace / --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn

dev = torch.device("cuda:0")
room = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth, K, T = syn.tsdf_frames(16, 240, 320, seed=5, bounds=room)
d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
f = OurFuser(None, 0.02, 3.0, bounds=room)
for _ in range(4):
    f.fuse_frames(d[:n], k[:n], t[:n], None)
torch.cuda.synchronize()
