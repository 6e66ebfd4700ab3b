#!/usr/bin/env python3
"""How long does the HOST need to issue one bench step (Python + ctypes + allocator), compared with the GPU time of the step?
If the two are close, the two-stream bench is launch-bound on the CPU side."""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

import bench


def main():
    dev = torch.device("cuda:0")
    inp, pyr, t, pyr_t = bench.build_inputs(dev, 1000)
    model = bench.build_model(dev)
    hint = {n: t[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}
    step = lambda: model.forward_from_features(pyr_t, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"],
                                               t["src_Ks"], t["cur_invK"], hint, return_mask=True)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    # GPU idle at the start of every measurement: the host runs ahead freely
    issue = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        issue.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 50
    print(f"host issue time per step: median {np.median(issue) * 1e3:.3f} ms (min {min(issue) * 1e3:.3f}); "
          f"step wall time, single stream: {gpu * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
