#!/usr/bin/env python3
"""Throughput of the small model's conv stack (CVEncoder + SkipDecoderRegression, 120x160, D=64, batch 1) when S independent
frames are in flight on S HIP streams: how much of the stack's single-stream time is latency another frame can fill."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch

import gpu_util as gu
from doubletake_amd.modules.networks import CVEncoder
from doubletake_amd.modules.networks_fast import SkipDecoderRegression
from doubletake_amd.utils import synthetic as syn


def main():
    h, w, D = 120, 160, 64
    enc = [64, 64, 128, 256, 512]
    cve = CVEncoder(D, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    dec = SkipDecoderRegression([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 1)
    gu.set_formula_weights(dec, 2, 0.7)
    vol = torch.from_numpy(syn.hash_normalish((1, D, h, w), 1)).to(gu.dev()).contiguous(memory_format=torch.channels_last)
    feats = [torch.from_numpy(f).to(gu.dev()).contiguous(memory_format=torch.channels_last)
             for f in syn.prior_pyramid(1, enc, 2 * h, 2 * w, 2)]

    def frame():
        return dec([feats[0]] + cve(vol, feats[1:]), with_depth=True)

    out = {}
    for S in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(S)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        n = 120
        for i in range(12):
            with torch.cuda.stream(streams[i % S]):
                frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                frame()
        torch.cuda.synchronize()
        out[f"streams_{S}_ms_per_frame"] = (time.perf_counter() - t0) / n * 1e3
    # the same with one hipGraph per stream (no host issue limit): the GPU-side overlap of S conv stacks
    for S in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        graphs = []
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(3):
                    frame()  # per-stream scratch / allocator warm-up before capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                keep = frame()
            graphs.append((g, keep))
        torch.cuda.synchronize()
        n = 240
        for i in range(12):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S][0].replay()
        torch.cuda.synchronize()
        out[f"graph_streams_{S}_ms_per_frame"] = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
