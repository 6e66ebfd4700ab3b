#!/usr/bin/env python3
"""Round 5: time-sliced vs space-sliced sharing of the chip between the volume kernel and the conv stacks (bench.py's workload,
no TSDF fusion, hwq 8).

  rr S          frame i entirely on stream i % S (bench.py's schedule): the persistent volume kernel (one workgroup per CU,
                each owning the whole CU) alternates with the other frames' conv stacks
  split S NV    volume stage of every frame on ONE dedicated stream with a compute-unit budget of NV (dt_cv_mlp_set_cu_budget:
                NV workgroups, back to back, frame after frame), conv stacks round-robin on S more streams behind per-frame
                events: the volume kernels keep NV CUs, the conv stacks of the frames in flight share the rest
  split+mask    the same with CU-masked streams (hipExtStreamCreateWithCUMask): the volume stream may only use CUs [0, NV),
                the conv streams only the others
Prints ms per frame / frames per second for each variant."""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

import bench
from doubletake_amd import _abi


def masked_stream(dev, lo, hi, total=256):
    hip = C.CDLL("libamdhip64.so.7")
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def main():
    dev = torch.device("cuda:0")
    L = _abi.lib()
    sets = []
    for j in range(4):
        _, _, t, pyr_t = bench.build_inputs(dev, 1000 + 97 * j)
        hint = {n: t[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}
        kw = dict(cur_feats=t["cur_feats"], src_feats=t["src_feats"], src_extrinsics=t["src_extrinsics"], src_poses=t["src_poses"],
                  src_Ks=t["src_Ks"], cur_invK=t["cur_invK"], min_depth=t["min_depth"], max_depth=t["max_depth"], return_mask=True,
                  cv_depth_hint_dict=hint)
        sets.append((kw, pyr_t))
    model = bench.build_model(dev).eval()
    for kw, pyr_t in sets:
        model.network_stage(pyr_t, *model.volume_stage(kw))
    torch.cuda.synchronize()
    n = int(os.environ.get("DT_FRAMES", "160"))

    def rr(S, streams=None):
        streams = streams or [torch.cuda.Stream(dev) for _ in range(S)]
        for i in range(n):
            kw, pyr_t = sets[i % len(sets)]
            with torch.cuda.stream(streams[i % S]):
                model.network_stage(pyr_t, *model.volume_stage(kw))

    def split(S, vstream, cstreams, depth=None):
        """volume stage of frame i on vstream; its conv stack on cstreams[i % S] behind an event.  `depth` bounds how far the
        volume stream may run ahead of the conv stacks (frames), so that cost volumes in flight stay bounded."""
        depth = depth or S
        conv_done = []
        for i in range(n):
            kw, pyr_t = sets[i % len(sets)]
            with torch.cuda.stream(vstream):
                if len(conv_done) >= depth:
                    vstream.wait_event(conv_done[i - depth])
                vol = model.volume_stage(kw)
                ev = torch.cuda.Event()
                ev.record(vstream)
            cs = cstreams[i % S]
            with torch.cuda.stream(cs):
                cs.wait_event(ev)
                model.network_stage(pyr_t, *vol)
                for v in vol:
                    if torch.is_tensor(v):
                        v.record_stream(cs)
                dn = torch.cuda.Event()
                dn.record(cs)
                conv_done.append(dn)

    def timed(label, fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{label:40s} {ms:.4f} ms/frame = {1e3 / ms:7.1f} f/s", flush=True)

    L.dt_cv_mlp_set_cu_budget(0)
    for S in (3, 4):
        timed(f"rr S={S}", lambda S=S: rr(S))
    vs = torch.cuda.Stream(dev)
    for S in (2, 3, 4):
        cst = [torch.cuda.Stream(dev) for _ in range(S)]
        for nv in (256, 224, 208, 192, 176, 160, 144, 128):
            L.dt_cv_mlp_set_cu_budget(nv if nv < 256 else 0)
            timed(f"split S={S} NV={nv}", lambda: split(S, vs, cst))
    L.dt_cv_mlp_set_cu_budget(0)
    if os.environ.get("DT_MASKS", "1") != "0":
        try:
            for nv in (208, 192, 176, 160, 144):
                vm = masked_stream(dev, 0, nv)
                for S in (3, 4):
                    cm = [masked_stream(dev, nv, 256) for _ in range(S)]
                    L.dt_cv_mlp_set_cu_budget(nv)
                    timed(f"split+mask S={S} NV={nv}", lambda: split(S, vm, cm))
        except Exception as e:  # noqa
            print("masked streams failed:", repr(e), flush=True)
    L.dt_cv_mlp_set_cu_budget(0)
    timed("rr S=4 (again)", lambda: rr(4))


if __name__ == "__main__":
    main()
