#!/usr/bin/env python3
"""Whole-model timing (mesh-hint volume -> CVEncoder -> decoder -> exp) at the BASELINE.json shapes other than the
bench's: cfg3 (full model, 512x384, batch 8), cfg4 (small, 512x384, batch 1), cfg5 (full, portrait 384x512, 96 planes,
batch 2), plus cfg2 with the full model.  hipEvents over back-to-back forward_from_features calls, one stream."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch

import gpu_util as gu
import test_model_fullsize_gpu as cases


def main():
    out = {}
    for name in ("cfg2_small", "cfg2_full", "cfg3_full_b8", "cfg3_small_b8", "cfg4_small", "cfg5_full_d96", "cfg5_small_d96"):
        model, inp, t, pyr = cases.build_case(name)
        pyr = [p.contiguous(memory_format=torch.channels_last) for p in pyr]
        call = lambda: model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"],
                                                   t["src_Ks"], t["cur_invK"], gu.hint_dict(t), return_mask=True)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        n = 20
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            call()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        bsz = cases.CASES[name][0]
        out[name] = {"ms_per_call": round(ms, 3), "batch": bsz, "frames_per_s": round(bsz / ms * 1e3, 1)}
        print(name, out[name], flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "time_configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
