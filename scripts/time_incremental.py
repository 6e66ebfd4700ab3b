#!/usr/bin/env python3
"""Per-frame latency of the incremental (online) mode through the product loop ``loops.run_incremental_scan`` (reference
test_incremental.py:172-372): frame t's hint needs the TSDF after t-1 -- marching cubes -> depth render -> TSDF weight
sampling -> matching encoder (new keyframe only; sources from the HBM feature cache) -> cost volume + CVEncoder + decoder
-> TSDF integrate.  640x480, 7 source views, 64 planes, batch 1.

Modes:  serial     feature cache, everything on one stream, eager launches (round 2's figure)
        lookahead  frame t+1's keyframe is encoded on a side stream while frame t runs (loops.matching_lookahead)
        graphs     the model part and the single-image encoder pass replayed from hipGraphs (model.enable_hip_graphs)
        programs   the same two replayed from launch programs recorded at the C ABI (model.enable_launch_programs)
Reported: wall-clock ms/frame over the scan (no host synchronisation inside the loop) and the FrameTimer's per-frame
hint_time / model_time (HIP events; model_time includes the matching encoder when it is not hidden)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
import torch.nn as nn

import bench
from doubletake_amd import loops
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn


class FixedPyramid(nn.Module):
    """Stand-in for the timm image-prior encoder (out of scope): returns the resident synthetic pyramid."""

    def __init__(self, pyr):
        super().__init__()
        self.pyr = pyr

    def forward(self, image):
        return self.pyr


def main():
    dev = torch.device("cuda:0")
    # DT_CONFIG=cfg4_small: the shape BASELINE.json configs[3] names for the incremental mode (512x384); default: the headline
    # frame size (640x480), where the model step alone is 1.69 ms on one stream
    cfg_name = os.environ.get("DT_CONFIG", "cfg2_small")
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[cfg_name])
    inp, pyr, t, pyr_t = bench.build_inputs(dev, 1000)
    model = bench.build_model(dev)
    model.encoder = FixedPyramid(pyr_t)
    H, W = bench.CFG["image_h"], bench.CFG["image_w"]
    H2, W2 = H // 2, W // 2
    bd = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    _, K, T = syn.tsdf_frames(64, H2, W2, seed=5, bounds=bd)
    Kt, Tt = torch.from_numpy(K).to(dev), torch.from_numpy(T).to(dev)
    invK, pose = torch.from_numpy(np.linalg.inv(K)).float().to(dev), torch.from_numpy(np.linalg.inv(T)).float().to(dev)
    n, k_src = int(os.environ.get("DT_FRAMES", "60")), bench.CFG["num_src"]
    images = torch.from_numpy(syn.hash_normalish((n + k_src, 3, H, W), 77)).to(dev)
    # relative poses / intrinsics of the bench frame (the geometry the volume kernel is timed on)
    eye = torch.eye(4, device=dev).view(1, 4, 4)
    Ks1 = torch.linalg.inv(t["cur_invK"])

    # per-frame dicts built once, outside the timed loops (what a dataloader hands over)
    src_cTw = (t["src_extrinsics"] @ Tt[0:1].unsqueeze(1)).contiguous()
    src_wTc = (pose[0:1].unsqueeze(1) @ t["src_poses"]).contiguous()
    frames_data = []
    for f in range(n):
        j = 0  # static camera so that the hint mesh stays in view
        cur = {"image_b3hw": images[f + k_src:f + k_src + 1], "frame_id_string": [f"{f + k_src:06d}"],
               "K_s0_b44": Kt[j:j + 1], "invK_s0_b44": invK[j:j + 1], "K_full_depth_b44": Kt[j:j + 1],
               "invK_s1_b44": t["cur_invK"], "cam_T_world_b44": Tt[j:j + 1], "world_T_cam_b44": pose[j:j + 1]}
        # source extrinsics such that cam_T_world_src @ world_T_cam_cur reproduces the bench frame's relative poses
        src = {"image_b3hw": images[f:f + k_src].flip(0).unsqueeze(0).contiguous(),
               "frame_id_string": [[f"{f + k_src - 1 - i:06d}"] for i in range(k_src)],
               "K_s1_b44": t["src_Ks"], "cam_T_world_b44": src_cTw, "world_T_cam_b44": src_wTc}
        frames_data.append((cur, src))
    torch.cuda.synchronize()

    def batches():
        for cur, src in frames_data:
            yield dict(cur), dict(src)

    def model_fn(cur, src):
        out = model("test", cur, src, return_mask=True)
        out["depth_pred_s0_b1hw"] = out["depth_pred_s0_b1hw"].clamp(1.0, 2.5)
        return out

    res = {"config": cfg_name, "image": [H, W]}
    modes = os.environ.get("DT_MODES", "serial,lookahead,graphs,programs,programs+lookahead,serial,programs,programs+lookahead").split(",")
    for mode in modes:
        model.matching_feature_cache.clear()
        model.use_feature_cache = True
        model.enable_hip_graphs("graphs" in mode)
        model.enable_launch_programs("programs" in mode)
        if "eagerenc" in mode:
            model._recorded_encoder = None  # (experiment: the model step from a program, the encoder pass eager)
        fuser = OurFuser(None, 0.04, 3.0, bounds=bd)
        timer = loops.FrameTimer()
        warm = 8
        it = batches()
        head = [next(it) for _ in range(warm)]
        look = loops.matching_lookahead(model) if "lookahead" in mode else None
        loops.run_incremental_scan(model_fn, fuser, head, (H2, W2), lookahead=look)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # (the loop restarts its frame counter: give it a non-empty TSDF by continuing with the same fuser; its first frame
        # therefore runs with an empty hint, which is what the first frame of a scan costs anyway)
        done = loops.run_incremental_scan(model_fn, fuser, it, (H2, W2), timer=timer, lookahead=look)
        issue = (time.perf_counter() - t0) / done * 1e3   # host time to ENQUEUE a frame (the loop never waits for the GPU)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / done * 1e3
        pf = timer.per_frame()
        res[mode] = {"frames": done, "wall_ms_per_frame": wall, "host_issue_ms_per_frame": issue, "frames_per_s": 1e3 / wall,
                     "hint_time_ms_median": float(np.median(pf["hint_time"][1:])),
                     "model_time_ms_median": float(np.median(pf["model_time"][1:])),
                     "matching_cache": dict(hits=model.matching_feature_cache.hits, misses=model.matching_feature_cache.misses)}
    _, verts, faces = fuser.get_mesh_pytorch3d()
    res["mesh_verts"], res["mesh_faces"] = int(verts.shape[0]), int(faces.shape[0])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
