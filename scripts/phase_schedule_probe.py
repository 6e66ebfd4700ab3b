#!/usr/bin/env python3
"""What ordering the two halves of the forward pass across HIP streams returns (bench.py's workload, no TSDF fusion).

round-robin: frame i entirely on stream i % S (bench.py --streams S).
phased:      groups of S frames: the S volume kernels one after the other (each owns every CU and all of its LDS), then
             the S conv stacks side by side on S streams; the next group's volumes start when this group's stacks are done.
Prints ms per frame for S = 1..6 in both orders."""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

import bench


def main():
    dev = torch.device("cuda:0")
    inp, pyr, t, pyr_t = bench.build_inputs(dev, 1000)
    model = bench.build_model(dev).eval()
    hint = {n: t[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}
    kw = dict(cur_feats=t["cur_feats"], src_feats=t["src_feats"], src_extrinsics=t["src_extrinsics"], src_poses=t["src_poses"],
              src_Ks=t["src_Ks"], cur_invK=t["cur_invK"], min_depth=t["min_depth"], max_depth=t["max_depth"], return_mask=True,
              cv_depth_hint_dict=hint)
    for _ in range(3):
        model.network_stage(pyr_t, *model.volume_stage(kw))
    torch.cuda.synchronize()
    n = int(os.environ.get("DT_FRAMES", "240"))

    def round_robin(S):
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                model.network_stage(pyr_t, *model.volume_stage(kw))

    def phased(S, chain_volumes=True):
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        conv_done = []
        for g in range(n // S):
            vols, vol_done = [], []
            for j in range(S):
                with torch.cuda.stream(streams[j]):
                    st = torch.cuda.current_stream()
                    for ev in conv_done:          # previous group's conv stacks
                        st.wait_event(ev)
                    if chain_volumes and vol_done:
                        st.wait_event(vol_done[-1])
                    vols.append(model.volume_stage(kw))
                    ev = torch.cuda.Event()
                    ev.record(st)
                    vol_done.append(ev)
            conv_done = []
            for j in range(S):
                with torch.cuda.stream(streams[j]):
                    st = torch.cuda.current_stream()
                    for ev in vol_done:
                        st.wait_event(ev)
                    model.network_stage(pyr_t, *vols[j])
                    ev = torch.cuda.Event()
                    ev.record(st)
                    conv_done.append(ev)

    for S in (1, 2, 3, 4, 6):
        row = [f"S={S}"]
        for name, fn in (("round-robin", round_robin), ("phased", phased)):
            fn(S)  # warm-up (allocator pools, scratch of the new streams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(S)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (n // S * S) * 1e3
            row.append(f"{name} {ms:.4f} ms = {1e3 / ms:.1f} f/s")
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    main()
