"""bench.py prints exactly one JSON line with the driver's contract fields (short run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0.3 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None or rf["traffic"] > 1e6
    dot = d["roofline_warp_match_dot"]
    assert dot["bound"] == "hbm" and dot["unit"] == "GB/s" and dot["peak"] == 8000.0
    # the bound statement: the kernel is vector-issue bound, the HBM clause is answered by a number
    bs = dot["bound_statement"]
    assert dot["bound_by"] == "valu" and bs["hbm_time_floor_ms"] < 0.005 < bs["valu_issue_floor_ms"] < dot["avg_launch_ms"]
    assert bs["max_hbm_frac_at_valu_floor"] < 0.40
    assert 0.0 < dot["valu_frac"] < 1.0 and 0.0 < dot["lds_frac"] < 1.0 and 0.0 < dot["batch8_512x384"]["lds_frac"] < 1.0
    rc = d["roofline_conv"]  # conv stack + heads next to the dominant kernel
    assert rc["bound"] == "mfma" and rc["peak"] == rf["peak"] and abs(rc["frac"] - rc["achieved"] / rc["peak"]) < 1e-9
    assert 7.0e10 < rc["direct_equivalent_flops_per_step"] < 8.5e10   # CVEncoder 37.0 G + SkipDecoder/heads 40.7 G
    assert 10 <= rc["launches"] <= 60 and rc["model_launches_per_step"] > rc["launches"]
    assert d["config"]["name"] == "cfg2_small"
    # the schedule is the product's (parallel.KeyframePipeline) and the model step is one recorded launch program per lane
    assert d["config"]["pipeline"] == "doubletake_amd.parallel.KeyframePipeline" and d["config"]["launch"].startswith("launch program")
    assert d["host_issue_ms_per_step"] < 0.6 * d["ms_per_step"]
    # several keyframes in flight by default; the dominant kernel's figure comes from the isolated (single-stream) leg of
    # the same run and says so, the in-region bracket stays beside it
    import bench as _bench

    assert d["config"]["streams"] == _bench.DEFAULT_STREAMS >= 3
    assert "single-stream leg" in rf["measured_in"] and rf["in_region_streams"] == _bench.DEFAULT_STREAMS
    assert abs(rf["avg_launch_ms"] - d["single_stream"]["dominant_kernel_avg_launch_ms"]) < 1e-9
    assert rf["in_region_avg_launch_ms"] >= 0.9 * rf["avg_launch_ms"]
    assert rf["avg_launch_ms"] <= d["ms_per_step"] * 1.05  # (a kernel of the step cannot last longer than the step)
    assert {leg["streams"] for leg in d["other_stream_counts"]} == {2, 3, 4} - {_bench.DEFAULT_STREAMS}
    assert all(k in d for k in _bench.LINE_KEYS)  # the same keys on every default-config line
    assert d["cpu_baseline"] is None and d["parity"] is None and set(d["null_because"]) == {"cpu_baseline", "parity"}
    enc = d["roofline_encoder"]
    assert enc["bound"] == "mfma" and 0.05 < enc["frac"] < 1.0 and enc["batched_1_plus_K"]["images"] == 8
    assert 6.0e10 < enc["batched_1_plus_K"]["direct_equivalent_flops"] < 7.0e10 and enc["single_image"]["launches"] >= 3
    e2e = d["end_to_end"]
    assert e2e["streams"] == 1 and e2e[f"streams_{_bench.DEFAULT_STREAMS}"]["cache_off"]["frames_per_s"] > 0


def test_bench_other_baseline_shape_prints_the_same_contract():
    """bench.py --config: the other BASELINE.json shapes through the same timed loop and JSON line (full model here)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--config", "cfg5_full_d96", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["config"]["name"] == "cfg5_full_d96" and "DepthDecoderPP" in d["config"]["workload"]
    assert d["config"]["frames_per_step_per_gpu"] == 2 and d["config"]["matching_resolution"] == [128, 96]
    assert abs(d["value"] - 2e3 / d["ms_per_step"]) / d["value"] < 1e-6
    assert d["cpu_baseline"] is None and "roofline_warp_match_dot" not in d
    assert d["roofline"]["bound"] == "mfma" and 0.3 < d["roofline"]["frac"] < 1.0
    assert d["roofline_conv"]["direct_equivalent_flops_per_step"] > 3e11   # 2 frames x (34.6 G + 292.1 G) x (96x128 / 120x160)


def test_bench_graph_mode_keeps_the_roofline_bracket():
    """bench.py --graph: the captured model step must still be cut around the volume kernel (the manager only cuts while an
    event hook is installed, so bench.py captures with a placeholder) -- otherwise the line would carry no kernel duration."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--graph", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-side-legs"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "NaN" not in lines[0], lines
    d = json.loads(lines[0])
    assert d["config"]["launch"].startswith("hipGraph") and d["config"]["input_sets"] == 1
    assert d["config"]["pipeline"] == "doubletake_amd.parallel.KeyframePipeline"
    assert 0.3 < d["roofline"]["frac"] < 1.0 and 0.3 < d["roofline"]["avg_launch_ms"] < 2.0
    assert d["other_stream_counts"] is None and "other_stream_counts" in d["null_because"]  # (side legs skipped)


def test_multi_stream_frame_pipelining_is_bit_identical():
    """parallel.KeyframePipeline (the schedule bench.py's timed loop submits its steps to): consecutive keyframes round-robin
    on S lanes with only the TSDF integrations chained by events.  Same frames, same order: depth maps and the fused volume
    must not change by a bit at 2, 3 and 4 keyframes in flight -- with eager launches and with one recorded launch program
    per lane (model.enable_launch_programs)."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import gpu_util as gu
    from doubletake_amd import parallel
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.utils import synthetic as syn

    dev = gu.dev()
    h, w, k, D = 48, 64, 3, 16
    model = DepthModelCVHint(4 * h, 4 * w, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=k + 1,
                             matching_encoder_type=None)
    gu.set_formula_weights(model, 5)
    model = model.to(dev)
    bd = dict(xmin=-2.0, xmax=2.0, ymin=-2.0, ymax=2.0, zmin=0.0, zmax=2.4)
    _, K, T = syn.tsdf_frames(9, 2 * h, 2 * w, seed=2, bounds=bd)
    Kt, Tt = torch.from_numpy(K).to(dev), torch.from_numpy(T).to(dev)
    frames = []
    for f in range(9):
        t = gu.to_dev(syn.volume_inputs(1, k, h, w, 16, 30 + f))
        pyr = [torch.from_numpy(p).to(dev).contiguous(memory_format=torch.channels_last)
               for p in syn.prior_pyramid(1, [64, 64, 128, 256, 512], 2 * h, 2 * w, 40 + f)]
        frames.append((t, pyr))

    def run(in_flight, programs):
        fuser = OurFuser(None, 0.04, 3.0, bounds=bd)
        sf = parallel.KeyframeShardFuser(dev, 1, 0, (2 * h, 2 * w), fuser=fuser)
        outs = []

        def keyframe(f):
            t, pyr = frames[f]
            out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"],
                                              t["src_Ks"], t["cur_invK"], gu.hint_dict(t), return_mask=True)
            outs.append(out["depth_pred_s0_b1hw"].clone())  # (program mode: the lane's output buffer is overwritten 'in_flight' steps later)
            return out["depth_pred_s0_b1hw"].clamp(0.8, 2.0), Kt[f:f + 1], Tt[f:f + 1]

        # (conv_plan: the same launch plans in every run -- the default "auto" selects the throughput plan with keyframes in
        #  flight, whose K splits sum in another fp32 order: a few 1e-6, tests/test_networks_gpu.py)
        with parallel.KeyframePipeline(dev, in_flight=in_flight, shard_fuser=sf, model=model, launch_programs=programs,
                                       conv_plan="latency") as pipe:
            assert (pipe.streams is None) == (in_flight == 1)
            for f in range(len(frames)):
                pipe.step(f, lambda f=f: keyframe(f))
            pipe.finish_pass()
        torch.cuda.synchronize(dev)
        if programs:
            info = model._recorded_forward.info()
            assert len(info) == in_flight and all(e["launches"] >= 20 and e["segments"] == 4 and e["patches"] >= 10 for e in info), info
        model.enable_launch_programs(False)
        t = fuser.tsdf_fuser_pred.tsdf
        return outs, t.tsdf_values.clone(), t.tsdf_weights.clone()

    d1, v1, w1 = run(1, False)
    assert (w1 > 0).sum().item() > 1000
    for programs in (False, True):
        for in_flight in ((2, 3, 4) if not programs else (1, 4)):
            d2, v2, w2 = run(in_flight, programs)
            for a, b in zip(d1, d2):
                assert torch.equal(a, b), (in_flight, programs)
            assert torch.equal(v1.view(torch.int16), v2.view(torch.int16)) and torch.equal(w1.view(torch.int16), w2.view(torch.int16)), \
                (in_flight, programs)


def test_forced_collective_line_carries_every_key_of_the_plain_line():
    """`bench.py --gpus 1 --force-dist` (the N > 1 code path on one GPU: RCCL group, per-step all_gather)
    prints a line with every key of the plain N = 1 line, plus the evidence that RCCL connected the ranks; slab-mode TSDF
    fusion runs through the same loop."""
    import bench as _bench

    base = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--force-dist"]
    for extra in ([], ["--tsdf-mode", "slab", "--tsdf-res", "0.02", "--no-side-legs"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, cwd=REPO)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert all(k in d for k in _bench.LINE_KEYS)
        assert d["config"]["ranks_seen"] == 1 and d["config"]["rccl_version"]
        assert d["roofline"]["frac"] > 0.3 and d["roofline_conv"]["launches"] >= 10 and d["single_stream"]["value"] > 0
        if extra:
            assert d["config"]["tsdf"] == {"mode": "slab", "voxel_m": 0.02} and "x-slab" in d["config"]["parallelism"]
        else:
            assert d["config"]["tsdf"] == {"mode": "replica", "voxel_m": 0.04}
