#!/usr/bin/env python3
"""cProfile of the HOST side of bench.py's model step (forward_from_features + fuse), 4 streams, no GPU waits in the loop:
where the ~1.0-1.2 ms per step that the host needs to enqueue ~50 launches goes.  python scripts/profile_step_host.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

import bench
from doubletake_amd.modules import conv_ops


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    sets = []
    for j in range(4):
        _, _, t, pyr_t = bench.build_inputs(dev, 1000 + 97 * j)
        sets.append((t, pyr_t, {k: t[k] for k in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}))
    model = bench.build_model(dev)
    conv_ops.set_plan_objective(conv_ops.PLAN_THROUGHPUT)
    streams = [torch.cuda.Stream(dev) for _ in range(4)]

    def step(i):
        t, p, h = sets[i % 4]
        with torch.cuda.stream(streams[i % 4]):
            return model.forward_from_features(p, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                               t["cur_invK"], h, return_mask=True)

    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    issue = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"unprofiled: host issue {issue:.4f} ms/step, wall {wall:.4f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print("\n".join(l[:170] for l in s.getvalue().splitlines()[:70]))


if __name__ == "__main__":
    main()
