#!/usr/bin/env python3
"""Throughput of the incremental (online) mode with several SCANS in flight on one GPU (loops.run_incremental_scans, round 6):
inside a scan frame t needs the TSDF after frame t-1, so one scan is a chain of latency-bound kernels; different scans are
independent and share the chip.  Same per-frame work as scripts/time_incremental.py ("serial" mode: hint from the TSDF ->
matching encoder on the new keyframe (feature cache) -> volume + CVEncoder + decoder -> fuse); S scans of N frames each, lanes =
1, 2, 3, 4.  Wall clock over all frames, no host synchronisation inside the loop.  Last entries ("batched", "batched_2_lanes"): the same scans in lock
step, ONE model call per turn on their collated keyframes (loops.IncrementalScanBatch), as one batch and as two batches on two lanes.

    DT_CONFIG=cfg4_small python scripts/time_incremental_scans.py     (DT_FRAMES=40 per scan, DT_SCANS=4, DT_LAUNCH=program|eager)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from doubletake_amd import hwqueues

hwqueues.ensure(4)
import numpy as np
import torch
import torch.nn as nn

import bench
from doubletake_amd import loops
from doubletake_amd.modules import conv_ops
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn


class FixedPyramid(nn.Module):
    def __init__(self, pyr):
        super().__init__()
        self.pyr = pyr
        self._by_batch = {1: pyr}

    def forward(self, image):
        b = image.shape[0]
        if b not in self._by_batch:  # (the batched mode evaluates k scans' keyframes in one call)
            self._by_batch[b] = [p.expand(b, -1, -1, -1).contiguous(memory_format=torch.channels_last) for p in self.pyr]
        return self._by_batch[b]


def main():
    dev = torch.device("cuda:0")
    cfg_name = os.environ.get("DT_CONFIG", "cfg4_small")
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[cfg_name])
    inp, pyr, t, pyr_t = bench.build_inputs(dev, 1000)
    model = bench.build_model(dev)
    model.encoder = FixedPyramid(pyr_t)
    model.use_feature_cache = True
    H, W = bench.CFG["image_h"], bench.CFG["image_w"]
    H2, W2 = H // 2, W // 2
    bd = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    _, K, T = syn.tsdf_frames(8, H2, W2, seed=5, bounds=bd)
    Kt, Tt = torch.from_numpy(K).to(dev), torch.from_numpy(T).to(dev)
    invK, pose = torch.from_numpy(np.linalg.inv(K)).float().to(dev), torch.from_numpy(np.linalg.inv(T)).float().to(dev)
    n, k_src, n_scans = int(os.environ.get("DT_FRAMES", "40")), bench.CFG["num_src"], int(os.environ.get("DT_SCANS", "4"))
    launch = os.environ.get("DT_LAUNCH", "program")
    images = torch.from_numpy(syn.hash_normalish((n + k_src, 3, H, W), 77)).to(dev)
    src_cTw = (t["src_extrinsics"] @ Tt[0:1].unsqueeze(1)).contiguous()
    src_wTc = (pose[0:1].unsqueeze(1) @ t["src_poses"]).contiguous()

    def scan_batches(s):
        for f in range(n):
            cur = {"image_b3hw": images[f + k_src:f + k_src + 1], "frame_id_string": [f"{f + k_src:06d}"], "scan_id_string": f"scan{s}",
                   "K_s0_b44": Kt[0:1], "invK_s0_b44": invK[0:1], "K_full_depth_b44": Kt[0:1],
                   "invK_s1_b44": t["cur_invK"], "cam_T_world_b44": Tt[0:1], "world_T_cam_b44": pose[0:1]}
            src = {"image_b3hw": images[f:f + k_src].flip(0).unsqueeze(0).contiguous(),
                   "frame_id_string": [[f"{f + k_src - 1 - i:06d}"] for i in range(k_src)],
                   "K_s1_b44": t["src_Ks"], "cam_T_world_b44": src_cTw, "world_T_cam_b44": src_wTc}
            yield cur, src

    def model_fn(cur, src):
        out = dict(model("test", cur, src, return_mask=True))
        out["depth_pred_s0_b1hw"] = out["depth_pred_s0_b1hw"].clamp(1.0, 2.5)
        return out

    res = {"config": cfg_name, "image": [H, W], "frames_per_scan": n, "scans": n_scans, "launch": launch}
    model.enable_launch_programs(launch == "program")
    for lanes in (1, 2, 3, 4, 1):
        if lanes > n_scans:
            continue
        conv_ops.set_plan_objective(conv_ops.PLAN_THROUGHPUT if lanes > 1 else conv_ops.PLAN_LATENCY)
        for timed in (False, True):  # a warm-up round (programs recorded per lane, allocator pools), then the timed one
            model.matching_feature_cache.clear()
            scans = [loops.IncrementalScan(model_fn, OurFuser(None, 0.04, 3.0, bounds=bd), scan_batches(s), (H2, W2))
                     for s in range(n_scans)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = loops.run_incremental_scans(scans, in_flight=lanes, device=dev)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        frames = sum(done)
        res[f"lanes_{lanes}" + ("_again" if f"lanes_{lanes}" in res else "")] = {
            "frames": frames, "ms_per_frame": wall / frames * 1e3, "frames_per_s": frames / wall,
            "conv_plan_mask": conv_ops.current_plan_objective()}
    # the same scans advanced in lock step, one model call per turn on the collated keyframes (loops.IncrementalScanBatch): one
    # batch of all scans on one stream, then two batches of half the scans each on two lanes
    for groups in (1, 2):
        if groups > n_scans:
            continue
        conv_ops.set_plan_objective(conv_ops.PLAN_THROUGHPUT if groups > 1 else conv_ops.PLAN_LATENCY)
        for timed in (False, True):
            model.matching_feature_cache.clear()
            scans = [loops.IncrementalScan(None, OurFuser(None, 0.04, 3.0, bounds=bd), scan_batches(s), (H2, W2)) for s in range(n_scans)]
            batches = [loops.IncrementalScanBatch(scans[g::groups], model_fn) for g in range(groups)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loops.run_incremental_scans(batches, in_flight=groups, device=dev)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        frames = sum(sc.frames for sc in scans)
        res["batched" if groups == 1 else f"batched_{groups}_lanes"] = {
            "frames": frames, "batch": n_scans // groups, "ms_per_frame": wall / frames * 1e3, "frames_per_s": frames / wall}
    conv_ops.set_plan_objective(conv_ops.PLAN_LATENCY)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
