import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn
from doubletake_amd.utils.pytorch3d_extras import marching_cubes_raw
dev = torch.device("cuda:0")
room = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
depth, K, T = syn.tsdf_frames(12, 240, 320, seed=5, bounds=room)
d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
f = OurFuser(None, 0.02, 3.0, bounds=room)
for i in range(12): f.fuse_frames(d[i:i+1], k[i:i+1], t[i:i+1], None)
tsdf = f.tsdf_fuser_pred.tsdf
for _ in range(10): marching_cubes_raw(tsdf.tsdf_values, tsdf.voxel_bitmap, 0.0)
torch.cuda.synchronize()
