"""CPU: the host logic of parallel.KeyframePipeline and doubletake_amd.hwqueues (no GPU: lanes are bookkeeping, the order of
steps, exchanges and integrations is what the multi-rank tests in test_parallel_gloo.py rely on)."""
import os
import warnings

import pytest
import torch

from doubletake_amd import hwqueues
from doubletake_amd import parallel as par


def test_hwqueues_ensure_and_check(monkeypatch):
    monkeypatch.delenv(hwqueues.VAR, raising=False)
    assert hwqueues.ensure(3) is None and hwqueues.VAR not in os.environ          # 3 lanes + the default stream fit 4 queues
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert hwqueues.check(3) is True
    with pytest.warns(UserWarning, match="hardware queues"):
        assert hwqueues.check(4) is False                                        # 5 streams on the runtime's default of 4
    assert hwqueues.ensure(4) == str(hwqueues.WANT) and os.environ[hwqueues.VAR] == "8"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert hwqueues.check(4) is True
    monkeypatch.setenv(hwqueues.VAR, "6")
    assert hwqueues.ensure(4) == "6"                                             # an explicit setting wins


def _fuser(log):
    return par.KeyframeShardFuser(torch.device("cpu"), 1, 0, (2, 3), fuse_fn=lambda d, K, T: log.append(float(d[0, 0, 0, 0])))


def test_pipeline_on_cpu_keeps_submission_order():
    log, ran = [], []
    pipe = par.KeyframePipeline(torch.device("cpu"), in_flight=4, shard_fuser=_fuser(log))
    assert pipe.streams is None and pipe.conv_plan_mask is None and [pipe.lane_of(i) for i in range(6)] == [0, 1, 2, 3, 0, 1]

    def frame(i):
        ran.append(i)
        return torch.full((1, 1, 2, 3), float(i)), torch.eye(4)[None], torch.eye(4)[None]

    for i in range(6):
        res, n = pipe.step(i, lambda i=i: frame(i))
        assert n == 1 and float(res[0][0, 0, 0, 0]) == i
    res, n = pipe.step(6, lambda: None)          # a rank without a batch in this step
    assert res is None and n == 0
    pipe.finish_pass()
    pipe.close()
    pipe.close()                                 # idempotent
    assert ran == list(range(6)) and log == [float(i) for i in range(6)]
    with pytest.raises(RuntimeError, match="after close"):
        pipe.step(7, lambda: None)


@pytest.mark.parametrize("in_flight", [1, 3, 4])
def test_run_sharded_pass_integrates_every_batch_in_order_at_any_in_flight(in_flight):
    log = []
    sizes = [2, 2, 1, 2, 1]

    def step_fn(i):
        d = torch.arange(sizes[i], dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 1, 2, 3) + 10.0 * i
        eye = torch.eye(4).repeat(sizes[i], 1, 1)
        return d.contiguous(), eye, eye

    sf = par.KeyframeShardFuser(torch.device("cpu"), 1, 0, (2, 3), fuse_fn=lambda d, K, T: log.extend(float(v) for v in d[:, 0, 0, 0]))
    n = par.run_sharded_pass(len(sizes), lambda i: sizes[i], step_fn, sf, in_flight=in_flight)
    assert n == sum(sizes) == sf.frames_fused
    assert log == [10.0 * i + j for i in range(len(sizes)) for j in range(sizes[i])]


def test_run_two_pass_without_a_final_fuser_runs_the_second_pass_through_the_pipeline():
    log, second = [], []
    sf = _fuser(log)
    first = lambda i: (torch.full((1, 1, 2, 3), float(i)), torch.eye(4)[None], torch.eye(4)[None])
    n1, n2 = par.run_two_pass(3, lambda i: 1, first, lambda i, state: second.append((i, state)), sf, None,
                              between_passes=lambda fuser: "mesh", in_flight=4)
    assert (n1, n2) == (3, 0) and log == [0.0, 1.0, 2.0] and second == [(0, "mesh"), (1, "mesh"), (2, "mesh")]


def test_incremental_scans_batched_collates_per_turn_and_shrinks_with_finished_scans(monkeypatch):
    """loops.run_incremental_scans_batched (round 6): one model call per turn on the collated frames of the scans still
    running -- tensors concatenated along the batch extent, frame ids concatenated per source view, scan ids as a list -- and
    every scan fuses its own element of the outputs; the call sequence of each scan equals its stand-alone sequence."""
    from doubletake_amd import loops

    fused = {}

    class Fuser:
        def __init__(self, name):
            self.name = name

        def fuse_frames(self, depth, K, T, c):
            fused.setdefault(self.name, []).append((float(depth.mean()), tuple(depth.shape)))

    hints = []
    monkeypatch.setattr(loops, "prepare_mesh_hint_fused", lambda f, cur, h, w: hints.append((f.name, cur["frame_id_string"][0])))
    monkeypatch.setattr(loops, "empty_hint", lambda cur, z: hints.append(("empty", cur["frame_id_string"][0])))

    def batches(n, base):
        for i in range(n):
            cur = {"cam_T_world_b44": torch.eye(4)[None], "K_full_depth_b44": torch.eye(4)[None], "v": torch.tensor([float(base + i)]),
                   "frame_id_string": [f"{base + i:03d}"], "scan_id_string": f"s{base}", "pyr": [torch.zeros(1, 2), torch.zeros(1, 3)],
                   "extra": {"t": torch.ones(1, 4)}}
            yield cur, {"frame_id_string": [[f"a{base + i}"], [f"b{base + i}"]], "image_b3hw": torch.zeros(1, 2, 3, 4, 4)}

    seen = []

    def model_fn(cur, src):
        k = cur["cam_T_world_b44"].shape[0]
        seen.append((cur["frame_id_string"], cur["scan_id_string"], src["frame_id_string"]))
        assert cur["pyr"][0].shape == (k, 2) and cur["pyr"][1].shape == (k, 3) and cur["extra"]["t"].shape == (k, 4)
        assert src["image_b3hw"].shape == (k, 2, 3, 4, 4)
        return {"depth_pred_s0_b1hw": cur["v"].view(-1, 1, 1, 1).expand(-1, 1, 2, 2).contiguous(), "note": "kept"}

    per_frame = []
    scans = [loops.IncrementalScan(None, Fuser(f"f{b}"), batches(n, b), (2, 2), on_frame=lambda i, c, o, b=b: per_frame.append((b, i, o["note"])))
             for n, b in ((2, 10), (3, 20), (1, 30))]
    assert loops.run_incremental_scans_batched(scans, model_fn) == [2, 3, 1]
    assert seen == [(["010", "020", "030"], ["s10", "s20", "s30"], [["a10", "a20", "a30"], ["b10", "b20", "b30"]]),
                    (["011", "021"], ["s10", "s20"], [["a11", "a21"], ["b11", "b21"]]),
                    (["022"], ["s20"], [["a22"], ["b22"]])]
    assert fused == {"f10": [(10.0, (1, 1, 2, 2)), (11.0, (1, 1, 2, 2))], "f20": [(20.0, (1, 1, 2, 2)), (21.0, (1, 1, 2, 2)), (22.0, (1, 1, 2, 2))],
                     "f30": [(30.0, (1, 1, 2, 2))]}
    assert hints[:3] == [("empty", "010"), ("empty", "020"), ("empty", "030")] and hints[3:] == [("f10", "011"), ("f20", "021"), ("f20", "022")]
    assert sorted(per_frame) == [(10, 0, "kept"), (10, 1, "kept"), (20, 0, "kept"), (20, 1, "kept"), (20, 2, "kept"), (30, 0, "kept")]
    with pytest.raises(ValueError):
        loops.run_incremental_scans_batched([loops.IncrementalScan(None, Fuser("x"), batches(1, 0), (2, 2), lookahead=lambda c, s: None)], model_fn)
