#!/usr/bin/env python3
"""is the CPU baseline of bench.py (oracle/torch_cpu_ref.py, kind "port") a fair stand-in for the reference?

Times, in THIS container (the only place /root/reference exists), on the same cores and thread count, the same frame
(cfg2: 640x480, 7 source views, 64 planes, batch 1, bench.py's synthetic inputs and formula weights):
  (i)  the imported reference: FeatureMeshHintVolumeManager (loop over planes) -> CVEncoder -> SkipDecoderRegression -> exp
  (ii) oracle/torch_cpu_ref.py: hint_volume_loop -> lowest_cost -> cv_encoder -> skip_decoder_regression -> exp
and compares their outputs.  Writes profiles/r5_cpu_ref_vs_port.json.  Not run on the GPU box (reads /root/reference).

    python tests/golden/cpu_ref_vs_port.py [--threads 8] [--repeats 3]

Lives under tests/ because it is checker infrastructure: it imports the oracle (and the reference), which only tests/, smoke()
and bench.py's cpu_baseline leg may do.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    import make_golden as mg
    import torch

    ref = mg.import_reference()
    torch.set_num_threads(args.threads)
    from doubletake_amd.utils import synthetic as syn
    from oracle import torch_cpu_ref as tref

    b, k, h, w, D, seed = 1, 7, 120, 160, 64, 1000
    enc = [64, 64, 128, 256, 512]
    inp = syn.volume_inputs(b, k, h, w, 16, seed)
    ti = {n: torch.from_numpy(v) for n, v in inp.items()}
    pyr = [torch.from_numpy(p) for p in syn.prior_pyramid(b, enc, 2 * h, 2 * w, seed + 50)]
    hint_names = ("depth_hint_b1hw", "depth_hint_mask_b1hw", "sampled_weights_b1hw")
    common = {n: ti[n] for n in ("cur_feats", "src_feats", "src_extrinsics", "src_poses", "src_Ks", "cur_invK", "min_depth", "max_depth")}
    N, NF = ref["networks"], ref["networks_fast"]
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        hv = ref["mesh_hint_volume"].FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1],
                                                                  matching_dim_size=16, num_source_views=k)
        mg.set_formula_weights(hv.mlp, seed + 1)
        mg.set_formula_weights(hv.hint_mlp, seed + 2)
        cve = N.CVEncoder(num_ch_cv=D, num_ch_enc=enc[1:], num_ch_outs=[64, 128, 256, 384])
        mg.set_formula_weights(cve, seed + 3)
        dec = NF.SkipDecoderRegression([enc[0]] + [64, 128, 256, 384])
        mg.set_formula_weights(dec, seed + 4)

    def run_reference():
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            vol, low, _, mask = hv(**common, cv_depth_hint_dict={n: ti[n].clone() for n in hint_names}, return_mask=True)
            t1 = time.perf_counter()
            out = dec([pyr[0]] + cve(vol, pyr[1:]))
            depth = {kk.replace("log_", ""): torch.exp(v) for kk, v in out.items() if kk.startswith("log_depth")}
            t2 = time.perf_counter()
        return t1 - t0, t2 - t1, vol, depth

    lin = lambda m: [(m.net[i].weight.detach(), m.net[i].bias.detach()) for i in (0, 2, 4)]
    sd = lambda m: {n: v.detach() for n, v in m.state_dict().items()}
    geo = (ti["cur_feats"], ti["src_feats"], ti["src_extrinsics"], ti["src_poses"], ti["src_Ks"], ti["cur_invK"], ti["min_depth"],
           ti["max_depth"], D, lin(hv.mlp))
    hint = {n: ti[n] for n in hint_names}

    def run_port():
        with torch.no_grad():
            t0 = time.perf_counter()
            vol, planes = tref.hint_volume_loop(*geo, hint=hint, hint_mlp=lin(hv.hint_mlp))
            t1 = time.perf_counter()
            tref.lowest_cost(vol, planes)
            out = tref.skip_decoder_regression([pyr[0]] + tref.cv_encoder(vol, pyr[1:], sd(cve)), sd(dec))
            depth = {kk.replace("log_", ""): torch.exp(v) for kk, v in out.items() if kk.startswith("log_depth")}
            t2 = time.perf_counter()
        return t1 - t0, t2 - t1, vol, depth

    rows = {"reference": [], "port": []}
    outs = {}
    for rep in range(args.repeats + 1):  # first pass = warm-up, interleaved so that both see the same machine state
        for name, fn in (("reference", run_reference), ("port", run_port)):
            tv, tr, vol, depth = fn()
            outs[name] = (vol, depth)
            if rep:
                rows[name].append((tv, tr))
            print(f"[{'warm' if not rep else rep}] {name:9s} volume {tv:6.2f} s  convs {tr:5.2f} s  frame {tv + tr:6.2f} s", flush=True)
    med = lambda xs: float(np.median(xs))
    res = {"config": "cfg2: 640x480, K=7, D=64, B=1; bench.py's synthetic frame (seed 1000) and formula weights", "threads": args.threads,
           "repeats": args.repeats, "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
           "logical_cpus": os.cpu_count(), "torch": torch.__version__}
    for name in rows:
        res[name] = {"volume_s": med([a for a, _ in rows[name]]), "convs_s": med([c for _, c in rows[name]]),
                     "frame_s": med([a + c for a, c in rows[name]])}
    res["port_over_reference"] = {kk: res["port"][kk] / res["reference"][kk] for kk in ("volume_s", "convs_s", "frame_s")}
    res["max_abs_diff"] = {"volume": float((outs["reference"][0] - outs["port"][0]).abs().max()),
                           **{kk: float((outs["reference"][1][kk] - outs["port"][1][kk]).abs().max()) for kk in outs["reference"][1]}}
    print(json.dumps(res, indent=1))
    with open(os.path.join(REPO, "profiles", "r5_cpu_ref_vs_port.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
