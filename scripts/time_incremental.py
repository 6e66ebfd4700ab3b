#!/usr/bin/env python3
"""Per-frame latency of the incremental (online) loop, where frame t's hint needs the TSDF after t-1
(reference test_incremental.py:172-372): marching cubes -> depth render -> TSDF weight sampling ->
matching encoder (new frame only, sources from the HBM feature cache) -> cost volume + CVEncoder + decoder
-> TSDF integrate.  640x480, 7 source views, 64 planes, batch 1."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

import bench
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn
from doubletake_amd.utils.rendering_utils import MeshDepthRenderer, empty_hint, prepare_mesh_hint, prepare_mesh_hint_fused


def main():
    dev = torch.device("cuda:0")
    inp, pyr, t, pyr_t = bench.build_inputs(dev, 1000)
    model = bench.build_model(dev)
    H2, W2 = bench.CFG["image_h"] // 2, bench.CFG["image_w"] // 2
    bd = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    _, K, T = syn.tsdf_frames(64, H2, W2, seed=5, bounds=bd)
    fuser = OurFuser(None, 0.04, 3.0, bounds=bd)
    renderer = MeshDepthRenderer(H2, W2)
    Kt, Tt = torch.from_numpy(K).to(dev), torch.from_numpy(T).to(dev)
    invK, pose = torch.from_numpy(np.linalg.inv(K)).float().to(dev), torch.from_numpy(np.linalg.inv(T)).float().to(dev)
    stages = {"hint": [], "matching": [], "model": [], "fuse": []}
    n = 40
    k_src = bench.CFG["num_src"]
    images = torch.from_numpy(syn.hash_normalish((n + k_src, 3, bench.CFG["image_h"], bench.CFG["image_w"]), 77)).to(dev)
    for f in range(n):
        j = 0  # static camera so the hint mesh is in view
        cur = {"K_s0_b44": Kt[j:j + 1], "invK_s0_b44": invK[j:j + 1], "cam_T_world_b44": Tt[j:j + 1], "world_T_cam_b44": pose[j:j + 1]}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        if f == 0:
            empty_hint(cur, torch.zeros(1, 1, H2, W2, device=dev))
        else:
            if os.environ.get("DT_HINT_COMPOSED"):
                prepare_mesh_hint(fuser, renderer, cur, H2, W2)
            else:
                prepare_mesh_hint_fused(fuser, cur, H2, W2)
        ev[1].record()
        # frame f + k_src is the new keyframe; its k_src predecessors are the sources (already cached after frame 0)
        cur_img = images[f + k_src:f + k_src + 1]
        src_img = images[f:f + k_src].flip(0).unsqueeze(0)
        m_cur, m_src = model.compute_matching_feats(cur_img, src_img, cur_ids=[f"{f + k_src:06d}"],
                                                    src_ids=[[f"{f + k_src - 1 - i:06d}"] for i in range(k_src)])
        ev[2].record()
        out = model.forward_from_features(pyr_t, m_cur, m_src, t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                          t["cur_invK"], cur, return_mask=True)
        ev[3].record()
        fuser.fuse_frames(out["depth_pred_s0_b1hw"].clamp(1.0, 2.5), cur["K_s0_b44"], cur["cam_T_world_b44"], None)
        ev[4].record()
        torch.cuda.synchronize()
        if f >= 5:
            stages["hint"].append(ev[0].elapsed_time(ev[1]))
            stages["matching"].append(ev[1].elapsed_time(ev[2]))
            stages["model"].append(ev[2].elapsed_time(ev[3]))
            stages["fuse"].append(ev[3].elapsed_time(ev[4]))
    res = {k: float(np.median(v)) for k, v in stages.items()}
    res["frame_ms"] = sum(res.values())
    res["frames_per_s"] = 1e3 / res["frame_ms"]
    _, verts, faces = fuser.get_mesh_pytorch3d()
    res["mesh_verts"], res["mesh_faces"] = int(verts.shape[0]), int(faces.shape[0])
    res["matching_cache"] = dict(hits=model.matching_feature_cache.hits, misses=model.matching_feature_cache.misses)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
