import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn
dev = torch.device("cuda:0")
for res in (0.04, 0.02):
    f = OurFuser(gt_path=None, fusion_resolution=res, max_fusion_depth=3.0)
    t = f.tsdf_fuser_pred.tsdf
    bd = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    depth, K, T = syn.tsdf_frames(8, 240, 320, seed=5, bounds=bd)
    d, k, tt = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
    for i in range(3):
        f.fuse_frames(d[i:i+1], k[i:i+1], tt[i:i+1], None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(8):
        f.fuse_frames(d[i:i+1], k[i:i+1], tt[i:i+1], None)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 8
    a.record(); _, v, fc = f.get_mesh_pytorch3d(); b.record(); torch.cuda.synchronize()
    print(res, tuple(t.tsdf_values.shape), f"integrate {ms:.3f} ms/frame, mesh {a.elapsed_time(b):.2f} ms, verts {v.shape[0]}, mem {torch.cuda.memory_allocated()/1e9:.2f} GB")
    del f, t
    torch.cuda.empty_cache()
