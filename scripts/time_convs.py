#!/usr/bin/env python3
"""hipEvent timing of the conv stacks at BASELINE configs[1] size (matching res 120x160, D=64)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import torch

import gpu_util as gu
from doubletake_amd.modules.networks import CVEncoder, DepthDecoderPP
from doubletake_amd.modules.networks_fast import SkipDecoderRegression
from doubletake_amd.utils import synthetic as syn


def timeit(fn, n=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    h, w, D = 120, 160, 64
    out = {}
    for name, enc, dec_cls, gf in (("small", [64, 64, 128, 256, 512], SkipDecoderRegression, 37.0 + 40.7),
                                   ("full", [24, 48, 64, 160, 256], DepthDecoderPP, 34.6 + 292.1)):
        cve = CVEncoder(D, enc[1:], [64, 128, 256, 384]).to(gu.dev())
        dec = dec_cls([enc[0], 64, 128, 256, 384]).to(gu.dev())
        gu.set_formula_weights(cve, 1)
        gu.set_formula_weights(dec, 2, 0.7)
        vol = torch.from_numpy(syn.hash_normalish((1, D, h, w), 1)).to(gu.dev()).contiguous(memory_format=torch.channels_last)
        feats = [torch.from_numpy(f).to(gu.dev()).contiguous(memory_format=torch.channels_last)
                 for f in syn.prior_pyramid(1, enc, 2 * h, 2 * w, 2)]
        t_enc = timeit(lambda: cve(vol, feats[1:]))
        cv = cve(vol, feats[1:])
        t_dec = timeit(lambda: dec([feats[0]] + cv))
        out[name] = dict(encoder_ms=t_enc, decoder_ms=t_dec, total_ms=t_enc + t_dec, tflops=gf / (t_enc + t_dec) * 1e-3 * 1e3 / 1e3)
        out[name]["tflops"] = gf * 1e9 / ((t_enc + t_dec) * 1e-3) / 1e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
