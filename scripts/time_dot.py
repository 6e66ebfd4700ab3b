#!/usr/bin/env python3
"""Timing + staging statistics of the dot-product volume kernel (csrc/cv_dot_lds.hip) at cfg2 (B=1, 120x160, K=7, D=64)
and cfg3 (B=8, 96x128).  `--once` runs a few launches only (for rocprofv3 --pmc passes)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch

from doubletake_amd.modules import cost_volume as cvmod
from doubletake_amd.utils import synthetic as syn


def run(b, k, h, w, D, seed, impl, n=30):
    dev = torch.device("cuda:0")
    t = {kk: torch.from_numpy(v).to(dev) for kk, v in syn.volume_inputs(b, k, h, w, 16, seed).items()}
    m = cvmod.CostVolumeManager(h, w, num_depth_bins=D).to(dev)
    evs = []

    def hook(tag):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)

    call = lambda: m(t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"], t["cur_invK"], t["min_depth"],
                     t["max_depth"])
    cvmod.CostVolumeManager._dot_impl = impl
    for _ in range(3):
        call()
    cvmod.CostVolumeManager._dot_event_hook = staticmethod(hook)
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    cvmod.CostVolumeManager._dot_event_hook = None
    cvmod.CostVolumeManager._dot_impl = "lds"
    ms = float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(0, len(evs), 2)]))
    st = m.last_dot_stats.tolist() if impl == "stats" else None
    return ms, st


def main():
    once = "--once" in sys.argv
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None  # one shape per process (per-shape PMC passes)
    out = {}
    for name, (b, k, h, w, D, seed) in {"cfg2": (1, 7, 120, 160, 64, 1000), "cfg3_b8": (8, 7, 96, 128, 64, 303)}.items():
        if only is not None and name != only:
            continue
        ms, _ = run(b, k, h, w, D, seed, "lds", 3 if once else 30)
        out[name] = {"lds_ms": ms}
        if not once:
            out[name]["direct_ms"], _ = run(b, k, h, w, D, seed, "direct")
            _, st = run(b, k, h, w, D, seed, "stats", 1)
            out[name]["units_staged_direct_empty_stray"] = st
            tiles = ((h + 15) // 16) * ((w + 15) // 16) * b
            out[name]["units_if_unsplit"] = tiles * k * ((D + 7) // 8)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
