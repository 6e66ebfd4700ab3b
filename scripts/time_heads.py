#!/usr/bin/env python3
"""Regression-head kernels one by one (HIP events, median of 50): the persistent kernel on the 240x320 map, the split kernel
on each coarse map, and the multi launch of the three coarse maps together (cfg2 shapes)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from doubletake_amd import _abi
from doubletake_amd.modules import mlp_pack


def main():
    dev = torch.device("cuda:0")
    L = _abi.lib()
    rng = np.random.default_rng(0)
    shapes = {"s0": (240 * 320, 64), "s1": (120 * 160, 64), "s2": (60 * 80, 128), "s3": (30 * 40, 256)}
    st = _abi.current_stream(dev)
    data = {}
    for n, (px, cin) in shapes.items():
        w = [rng.standard_normal(s).astype(np.float32) * 0.1 for s in ((128, cin, 1, 1), (128,), (128, 128, 1, 1), (128,), (1, 128, 1, 1), (1,))]
        pk = {k: torch.from_numpy(v).to(dev) for k, v in mlp_pack.pack_head_mlp(*w).items()}
        x = torch.randn(px, cin, device=dev)
        data[n] = (x, pk, torch.empty(px, device=dev), torch.empty(px, device=dev), px, cin)

    def single(n):
        x, pk, o, oe, px, cin = data[n]
        return lambda: _abi.check(L.dt_head_mlp_f32(_abi.ptr(x), _abi.ptr(pk["wa"]), _abi.ptr(pk["wb"]), _abi.ptr(pk["tail"]),
                                                    _abi.ptr(o), _abi.ptr(oe), px, cin, st), "head")

    def multi(names):
        k = len(names)
        tab = lambda vals: (C.c_void_p * k)(*[_abi.ptr(v) for v in vals])
        d = [data[n] for n in names]
        pixels = (C.c_int64 * k)(*[v[4] for v in d])
        cin = (C.c_int * k)(*[v[5] for v in d])
        args = (k, tab([v[0] for v in d]), tab([v[1]["wa"] for v in d]), tab([v[1]["wb"] for v in d]),
                tab([v[1]["tail"] for v in d]), tab([v[2] for v in d]), tab([v[3] for v in d]), pixels, cin, st)
        return lambda: _abi.check(L.dt_head_mlp_multi_f32(*args), "multi")

    def timeit(fn, reps=50):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return float(np.median(ts))

    for n in ("s0", "s1", "s2", "s3"):
        px, cin = shapes[n]
        gf = 2.0 * px * (cin * 128 + 128 * 128 + 128) / 1e9
        t = timeit(single(n))
        print(f"{n}: {px:6d} px, cin {cin:3d}: {t:7.2f} us  ({gf:.2f} GF -> {gf / t * 1e3:.1f} TF)")
    for names in (("s1", "s2", "s3"), ("s1", "s2"), ("s2", "s3"), ("s1",)):
        print(f"multi {'+'.join(names)}: {timeit(multi(names)):7.2f} us")


if __name__ == "__main__":
    main()
