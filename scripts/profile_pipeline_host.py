#!/usr/bin/env python3
"""cProfile of the HOST side of one bench.py step through the product pipeline (parallel.KeyframePipeline, 4 lanes, launch
programs, TSDF fuse): where the host time per step goes once the model step is one C call.  The back-pressure is switched off
(max_lead=0) so that the profile holds issue work, not waiting.  python scripts/profile_pipeline_host.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from doubletake_amd import hwqueues

hwqueues.ensure(4)
import torch

import bench
from doubletake_amd import parallel
from doubletake_amd.tools.fusers_helper import OurFuser
from doubletake_amd.utils import synthetic as syn


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    sets = []
    for j in range(4):
        _, _, t, pyr_t = bench.build_inputs(dev, 1000 + 97 * j)
        sets.append((t, pyr_t, {k: t[k] for k in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}))
    model = bench.build_model(dev)
    room = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    H2, W2 = 240, 320
    fuser = parallel.KeyframeShardFuser(dev, 1, 0, (H2, W2), fuser=OurFuser(None, 0.04, 3.0, bounds=room))
    _, Kp, Tp = syn.tsdf_frames(64, H2, W2, seed=5, bounds=room)
    K16, T16 = torch.from_numpy(Kp).to(dev).half(), torch.from_numpy(Tp).to(dev).half()
    pipe = parallel.KeyframePipeline(dev, in_flight=4, shard_fuser=fuser, model=model, launch_programs=True, max_lead=0)

    def keyframe(i):
        t, p, h = sets[i % 4]
        out = model.forward_from_features(p, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                          t["cur_invK"], h, return_mask=True)
        j = i % 64
        return out["depth_pred_s0_b1hw"], K16[j:j + 1], T16[j:j + 1]

    def step(i):
        pipe.step(i, lambda: keyframe(i))

    for i in range(24):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    issue = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    print(f"unprofiled: host issue {issue:.4f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(40)
        print("\n".join(l[:160] for l in s.getvalue().splitlines()[:62]))
    pipe.close()


if __name__ == "__main__":
    main()
