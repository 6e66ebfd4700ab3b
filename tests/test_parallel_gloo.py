"""world_size-2 CPU (gloo) tests of the multi-GPU modes of doubletake_amd/parallel.py: the keyframe-shard exchange
(canonical order, payload integrity, ragged schedules), the sharded two-pass loop (replicas bit-identical to a serial
run, checked by integrating with the numpy TSDF oracle), and the scene-sharded incremental loop with the
size-then-padded gather of finished TSDFs.  The collective logic is device independent; the HIP fuser is replaced by a
recording callback or by the oracle."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from doubletake_amd import parallel as par
from doubletake_amd.utils import synthetic as syn

BD = dict(xmin=-1.28, xmax=1.28, ymin=-1.12, ymax=1.12, zmin=0.0, zmax=2.24)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _spawn(fn, world, *args):
    mp.spawn(fn, args=(world, _free_port(), *args), nprocs=world, join=True)


# ---- per-step exchange ---------------------------------------------------------------------------
def _exchange_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    h, w = 6, 8
    log = []
    fuser = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (h, w),
                                   fuse_fn=lambda d, K, T: log.append((d.clone(), K.clone(), T.clone())))
    _, K, T = syn.tsdf_frames(16, h, w, seed=5, bounds=BD)
    for step in range(3):
        g = step * world + rank
        depth = torch.full((1, 1, h, w), float(10 * step + rank + 1)) + torch.arange(w).float() * 0.01
        n = fuser.exchange_and_fuse(depth, torch.from_numpy(K[g:g + 1]), torch.from_numpy(T[g:g + 1]))
        assert n == world
    torch.save(log, os.path.join(out_dir, f"log{rank}.pt"))
    dist.destroy_process_group()


def test_exchange_two_ranks(tmp_path):
    world = 2
    _spawn(_exchange_worker, world, str(tmp_path))
    logs = [torch.load(os.path.join(tmp_path, f"log{r}.pt")) for r in range(world)]
    assert len(logs[0]) == len(logs[1]) == 3
    _, K, T = syn.tsdf_frames(16, 6, 8, seed=5, bounds=BD)
    for step in range(3):
        d0, K0, T0 = logs[0][step]
        d1, K1, T1 = logs[1][step]
        # every replica sees the same frames in the same (rank-major) order
        assert torch.equal(d0, d1) and torch.equal(K0, K1) and torch.equal(T0, T1)
        assert d0.dtype == torch.float16 and tuple(d0.shape) == (2, 1, 6, 8)
        for r in range(world):
            want = (torch.full((6, 8), float(10 * step + r + 1)) + torch.arange(8).float() * 0.01).half()
            assert torch.equal(d0[r, 0], want)
            # the caller's cameras travel with the depth (cast to half exactly as OurFuser.fuse_frames does)
            np.testing.assert_array_equal(K0[r].numpy(), K[step * world + r].astype(np.float16))
            np.testing.assert_array_equal(T0[r].numpy(), T[step * world + r].astype(np.float16))


def test_pack_roundtrip_and_sharding():
    d = torch.rand(3, 1, 5, 7)
    K = torch.rand(3, 4, 4)
    T = torch.rand(3, 4, 4)
    dd, KK, TT = par.unpack_update(par.pack_update(d, K, T), 5, 7)
    assert torch.equal(dd, d.half()) and torch.equal(KK, K.half()) and torch.equal(TT, T.half())
    shards = [par.shard_keyframes(11, 4, r) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(11))
    assert shards[1] == [1, 5, 9]


def test_single_rank_needs_no_process_group():
    log = []
    f = par.KeyframeShardFuser(torch.device("cpu"), 1, 0, (4, 5), fuse_fn=lambda d, K, T: log.append(d), upsample_to=(8, 10))
    d = torch.arange(20.0).view(1, 1, 4, 5)
    assert f.exchange_and_fuse(d, torch.eye(4)[None], torch.eye(4)[None]) == 1
    assert tuple(log[0].shape) == (1, 1, 8, 10) and torch.equal(log[0][0, 0, ::2, ::2], d[0, 0])
    assert f.exchange_and_fuse(None, None, None) == 0


# ---- sharded two-pass loop -----------------------------------------------------------------------------
H, W = 24, 32
NB = 5                      # keyframe batches of the "scan": sizes 2,2,2,2,1 (ragged tail, odd batch count)
SIZES = [2, 2, 2, 2, 1]


def _scan_frames():
    depth, K, T = syn.tsdf_frames(sum(SIZES), H, W, seed=9, bounds=BD)
    return (depth * np.float32(0.6)).astype(np.float32), K, T


def _batch(i):
    depth, K, T = _scan_frames()
    s = sum(SIZES[:i])
    sl = slice(s, s + SIZES[i])
    return torch.from_numpy(depth[sl]), torch.from_numpy(K[sl]), torch.from_numpy(T[sl])


class _OracleFuser:
    """fuse_frames on the numpy TSDF oracle: lets a CPU test check replica bits."""

    def __init__(self, voxel):
        from oracle import tsdf_ref

        self.ref = tsdf_ref
        self.vol = tsdf_ref.TSDFVolume(BD, voxel)
        self.order = []

    def fuse_frames(self, d, K, T, color=None):
        assert d.dtype == torch.float16 and K.dtype == torch.float16
        for j in range(d.shape[0]):
            self.ref.integrate(self.vol, d[j, 0].numpy(), K[j].numpy(), T[j].numpy(), max_depth=3.0)
            self.order.append(float(d[j, 0, 0, 0]))


def _two_pass_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    hint = _OracleFuser(0.08)
    final = _OracleFuser(0.08)
    calls = {"first": [], "second": [], "between": 0}

    def first(i):
        calls["first"].append(i)
        return _batch(i)

    def between(f):
        assert f is hint
        calls["between"] += 1
        return {"w": hint.vol.weights.copy()}

    def second(i, state):
        calls["second"].append(i)
        assert state["w"].sum() > 0
        d, K, T = _batch(i)
        return d * 1.01, K, T

    sf_hint = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=hint)
    sf_final = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=final)
    n1, n2 = par.run_two_pass(NB, lambda i: SIZES[i], first, second, sf_hint, sf_final, between_passes=between)
    assert (n1, n2) == (sum(SIZES), sum(SIZES)) and calls["between"] == 1
    assert calls["first"] == calls["second"] == par.shard_keyframes(NB, world, rank)
    np.savez(os.path.join(out_dir, f"vol{rank}.npz"), hv=hint.vol.values, hw=hint.vol.weights, fv=final.vol.values,
             fw=final.vol.weights, order=np.array(hint.order))
    dist.destroy_process_group()


def test_two_pass_replicas_equal_serial_run(tmp_path):
    world = 2
    _spawn(_two_pass_worker, world, str(tmp_path))
    vols = [np.load(os.path.join(tmp_path, f"vol{r}.npz")) for r in range(world)]
    # serial single-process run over the same batches
    hint, final = _OracleFuser(0.08), _OracleFuser(0.08)
    for i in range(NB):
        d, K, T = _batch(i)
        hint.fuse_frames(d.half(), K.half(), T.half())
        final.fuse_frames((d * 1.01).half(), K.half(), T.half())
    assert (hint.vol.weights > 0).sum() > 500
    for v in vols:
        np.testing.assert_array_equal(v["order"], np.array(hint.order))  # canonical = serial batch order
        for got, want in ((v["hv"], hint.vol.values), (v["hw"], hint.vol.weights), (v["fv"], final.vol.values),
                          (v["fw"], final.vol.weights)):
            np.testing.assert_array_equal(got.view(np.uint16), want.view(np.uint16))  # bit-identical replicas


# ---- voxel-slab fusion (large final volumes): every rank integrates all gathered frames into its x-slab only ----------
class _OracleSlabFuser(_OracleFuser):
    """The numpy oracle restricted to an x-slab: integrates, then puts back every plane outside the slab -- what a kernel
    that only visits the slab leaves behind."""

    def set_slab(self, x0, x1):
        self.x0, self.x1 = x0, x1

    def slab_arrays(self):
        return [torch.from_numpy(self.vol.values.view(np.int16)), torch.from_numpy(self.vol.weights.view(np.int16))]

    def fuse_frames(self, d, K, T, color=None):
        keep_v, keep_w = self.vol.values.copy(), self.vol.weights.copy()
        super().fuse_frames(d, K, T, color)
        for arr, keep in ((self.vol.values, keep_v), (self.vol.weights, keep_w)):
            arr[: self.x0] = keep[: self.x0]
            arr[self.x1:] = keep[self.x1:]


def _slab_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    final = _OracleSlabFuser(0.08)
    sf = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=final, mode="slab")
    X = final.vol.values.shape[0]
    assert sf.slab == par.slab_bounds(X, world, rank) and sf.slab[1] - sf.slab[0] in (X // world, (X + world - 1) // world)
    n = par.run_sharded_pass(NB, lambda i: SIZES[i], _batch, sf)
    assert n == sum(SIZES) and sf.slabs_current
    # a second pass continues on the gathered volume (order-dependent running mean: the slabs must have been completed)
    n = par.run_sharded_pass(NB, lambda i: SIZES[i], lambda i: tuple(t * 1.02 if j == 0 else t for j, t in enumerate(_batch(i))), sf)
    np.savez(os.path.join(out_dir, f"slab{rank}.npz"), v=final.vol.values, w=final.vol.weights)
    dist.destroy_process_group()


def test_slab_fusion_assembles_the_serial_volume_bit_for_bit(tmp_path):
    """VERDICT r3 item 5 (alternative): x-slab sharding of the final volume; the gathered volume of every
    rank equals a serial run over the same batches (two passes, 9 keyframes each, ragged tail)."""
    world = 2
    _spawn(_slab_worker, world, str(tmp_path))
    serial = _OracleFuser(0.08)
    for scale in (1.0, 1.02):
        for i in range(NB):
            d, K, T = _batch(i)
            serial.fuse_frames((d * scale).half(), K.half(), T.half())
    assert (serial.vol.weights > 0).sum() > 500
    X = serial.vol.values.shape[0]
    # the test is only meaningful if both slabs were written to
    x0, x1 = par.slab_bounds(X, world, 0)
    assert (serial.vol.weights[x0:x1] > 0).any() and (serial.vol.weights[x1:] > 0).any()
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"slab{r}.npz"))
        np.testing.assert_array_equal(got["v"].view(np.uint16), serial.vol.values.view(np.uint16))
        np.testing.assert_array_equal(got["w"].view(np.uint16), serial.vol.weights.view(np.uint16))


def test_slab_bounds_cover_the_volume():
    for X, world in ((400, 8), (200, 8), (56, 3), (8, 16), (33, 2)):
        spans = [par.slab_bounds(X, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == X
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all(x0 <= x1 for x0, x1 in spans)


# ---- bench.py's step order (several frames in flight, round-robin over the streams) against the exchange ---------------------------
def _bench_default_streams():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return int(mod.DEFAULT_STREAMS)


BENCH_STREAMS = _bench_default_streams()  # keyframes in flight in bench.py's default run (round 5: 4)


def _bench_order_worker(rank, world, port, out_dir):
    """bench.py issues step i from HIP stream i % S (S = bench.DEFAULT_STREAMS); the collective of a step is enqueued from whichever stream runs it.
    What has to hold for RCCL is that every rank issues the SAME sequence of collectives (same order, same shapes) and
    integrates in the same canonical order -- the streams only change where the launch is enqueued.  Replayed here with the
    stream of a step recorded instead of used; the schedule ends ragged (7 keyframes on 2 ranks)."""
    _init(rank, world, port)
    h, w = 6, 8
    log, calls = [], []
    real = dist.all_gather_into_tensor

    def spy(out, inp, *a, **kw):
        calls.append((tuple(out.shape), tuple(inp.shape), str(inp.dtype)))
        return real(out, inp, *a, **kw)

    par.dist.all_gather_into_tensor = spy
    fuser = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (h, w),
                                   fuse_fn=lambda d, K, T: log.extend(float(v) for v in d[:, 0, 0, 0]))
    nframes = 7
    _, K, T = syn.tsdf_frames(8, h, w, seed=5, bounds=BD)
    issue = []
    for step in range((nframes + world - 1) // world):
        stream = step % BENCH_STREAMS           # bench.py: streams[frame_idx % len(streams)]
        g = step * world + rank                 # bench.py: (frame_idx * world + rank) -> camera / keyframe index
        counts = [1 if step * world + r < nframes else 0 for r in range(world)]
        have = g < nframes
        depth = torch.full((1, 1, h, w), float(g + 1)) if have else None
        fuser.exchange_and_fuse(depth, torch.from_numpy(K[g:g + 1]) if have else None, torch.from_numpy(T[g:g + 1]) if have else None,
                                counts=counts, rows=1)
        issue.append((step, stream, tuple(counts)))
    par.dist.all_gather_into_tensor = real
    torch.save({"log": log, "calls": calls, "issue": issue}, os.path.join(out_dir, f"bo{rank}.pt"))
    dist.destroy_process_group()


def test_bench_step_order_issues_identical_collectives_on_every_rank(tmp_path):
    world = 2
    _spawn(_bench_order_worker, world, str(tmp_path))
    r0, r1 = (torch.load(os.path.join(tmp_path, f"bo{r}.pt")) for r in range(world))
    assert r0["calls"] == r1["calls"] and len(r0["calls"]) == 4          # one collective per step, same shapes, same order
    assert r0["issue"] == r1["issue"] and [s for _, s, _ in r0["issue"]] == [i % BENCH_STREAMS for i in range(4)]
    assert r0["issue"][-1][2] == (1, 0)                                   # ragged last step: rank 1 has no keyframe
    assert r0["log"] == r1["log"] == [float(i + 1) for i in range(7)]     # canonical (= serial) integration order on both


def _bench_line_worker(rank, world, port, out_dir):
    """A bench.py-shaped step loop on two gloo ranks (the product kernels need a GPU; the loop, the exchange, the max-over-ranks
    timing and the JSON assembly do not): rank 0 prints the line bench.py's `complete_line` would."""
    import json
    import time

    _init(rank, world, port)
    spec_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(spec_dir, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h, w, steps = 6, 8, 5
    fused = []
    fuser = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (h, w), fuse_fn=lambda d, K, T: fused.append(int(d.shape[0])))
    _, K, T = syn.tsdf_frames(steps * world, h, w, seed=5, bounds=BD)
    dist.barrier()
    t0 = time.perf_counter()
    for step in range(steps):
        g = step * world + rank
        fuser.exchange_and_fuse(torch.full((1, 1, h, w), float(g + 1)), torch.from_numpy(K[g:g + 1]), torch.from_numpy(T[g:g + 1]))
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    if rank == 0:
        line = {"metric": "depth frames/sec (640x480, 7 src views, 64 planes)", "value": steps * world / elapsed, "unit": "frames/s",
                "n_gpus": world, "steps": steps, "warmup": 0, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "step loop of bench.py on gloo", "ranks_seen": int(dist.get_world_size()),
                           "frames_fused_per_step": fused[0]},
                "roofline": {"bound": "mfma"}, "roofline_conv": {"bound": "mfma"}, "single_stream": {"value": 1.0}}
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            f.write(json.dumps(bench.complete_line(line, world)))
    dist.destroy_process_group()


def test_two_rank_bench_line_has_every_key_of_the_single_gpu_line(tmp_path):
    """the line a 2-rank run prints carries every key of the N = 1 line (objects that are measured at N = 1
    only are null with the reason), reports whole-job frames/s over the max-over-ranks time, and shows how many ranks met."""
    import json

    world = 2
    _spawn(_bench_line_worker, world, str(tmp_path))
    d = json.loads(open(os.path.join(tmp_path, "line.json")).read())
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_for_tests2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert all(k in d for k in bench.LINE_KEYS)
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["frames_fused_per_step"] == 2
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-9
    assert d["cpu_baseline"] is None and "N=1 only" in d["null_because"]["cpu_baseline"]
    assert d["roofline"] is not None and d["roofline_conv"] is not None and d["single_stream"] is not None


# ---- revisit flow (test_revisit.py:104-260): first pass over a PREVIOUS scan, second pass over the new one -------------
FIRST_SIZES = [2, 2, 1]          # the previous scan has its own (shorter, ragged) batch schedule
REVISIT_SHIFT = np.array([[1, 0, 0, 0.10], [0, 1, 0, -0.05], [0, 0, 1, 0.0], [0, 0, 0, 1]], dtype=np.float32)


def _first_scan_batch(i):
    """Previous scan: the same room seen from cameras expressed in ITS world frame = REVISIT_SHIFT @ new world."""
    depth, K, T = _scan_frames()
    s = sum(FIRST_SIZES[:i])
    sl = slice(s, s + FIRST_SIZES[i])
    T_first = T[sl] @ np.linalg.inv(REVISIT_SHIFT)[None]  # cam_T_firstworld = cam_T_world @ world_T_firstworld
    return torch.from_numpy(depth[sl] * np.float32(0.97)), torch.from_numpy(K[sl]), torch.from_numpy(T_first.astype(np.float32))


def _revisit_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    from doubletake_amd import loops

    hint, final = _OracleFuser(0.08), _OracleFuser(0.08)
    seen = {"first": [], "second": [], "T": []}

    def model_fn(cur, src):  # stands in for the network: returns the batch's depth
        return {"depth_pred_s0_b1hw": cur["depth"], "overall_mask_bhw": None}

    def as_batch(d, K, T):
        return {"depth": d, "K_full_depth_b44": K, "cam_T_world_b44": T, "world_T_cam_b44": torch.linalg.inv(T)}, {}

    def load_first(i):
        seen["first"].append(i)
        return as_batch(*_first_scan_batch(i))

    def load_new(i):
        seen["second"].append(i)
        return as_batch(*_batch(i))

    first, between, second = loops.revisit_fns(model_fn, load_first, load_new, torch.from_numpy(REVISIT_SHIFT)[None], (H, W))
    # the GPU hint maps are covered by tests/test_drivers_gpu.py; here the step functions are driven through the sharded
    # schedule with the hint preparation swapped for a recorder of the pose it would render from
    def second_cpu(i, state):
        cur, src = load_new(i)
        pose = torch.from_numpy(REVISIT_SHIFT)[None] @ cur["world_T_cam_b44"]
        seen["T"].append(pose.clone())
        assert state["w"].sum() > 0
        return cur["depth"] * 1.01, cur["K_full_depth_b44"], cur["cam_T_world_b44"]

    sf_hint = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=hint)
    sf_final = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=final)
    n1, n2 = par.run_two_pass(NB, lambda i: SIZES[i], first, second_cpu, sf_hint, sf_final,
                              between_passes=lambda f: {"w": f.vol.weights.copy()}, num_first_batches=len(FIRST_SIZES),
                              first_batch_size_of=lambda i: FIRST_SIZES[i])
    assert (n1, n2) == (sum(FIRST_SIZES), sum(SIZES))
    assert seen["first"] == par.shard_keyframes(len(FIRST_SIZES), world, rank)
    assert seen["second"] == par.shard_keyframes(NB, world, rank)
    assert callable(second) and callable(between)
    np.savez(os.path.join(out_dir, f"rv{rank}.npz"), hv=hint.vol.values, hw=hint.vol.weights, fv=final.vol.values,
             fw=final.vol.weights)
    dist.destroy_process_group()


def test_revisit_flow_two_ranks_equals_serial(tmp_path):
    world = 2
    _spawn(_revisit_worker, world, str(tmp_path))
    vols = [np.load(os.path.join(tmp_path, f"rv{r}.npz")) for r in range(world)]
    hint, final = _OracleFuser(0.08), _OracleFuser(0.08)
    for i in range(len(FIRST_SIZES)):
        d, K, T = _first_scan_batch(i)
        hint.fuse_frames(d.half(), K.half(), T.half())
    for i in range(NB):
        d, K, T = _batch(i)
        final.fuse_frames((d * 1.01).half(), K.half(), T.half())
    assert (hint.vol.weights > 0).sum() > 300
    for v in vols:
        for got, want in ((v["hv"], hint.vol.values), (v["hw"], hint.vol.weights), (v["fv"], final.vol.values),
                          (v["fw"], final.vol.weights)):
            np.testing.assert_array_equal(got.view(np.uint16), want.view(np.uint16))


# ---- scene-sharded incremental mode ------------------------------------------------------------------------
SCENE_FRAMES = [40, 10, 25, 12, 5]   # 5 scans, 2 ranks -> LPT: rank0 [0, 4], rank1 [2, 3, 1]? checked below
SCENE_DIMS = [(8, 8, 8), (16, 8, 8), (8, 16, 8), (8, 8, 24), (24, 8, 8)]


class _FakeTsdf:
    def __init__(self, scene):
        g = torch.Generator().manual_seed(100 + scene)
        self.tsdf_values = (torch.rand(SCENE_DIMS[scene], generator=g) * 2 - 1).half()
        self.tsdf_weights = torch.rand(SCENE_DIMS[scene], generator=g).half()
        self.origin_f32 = np.array([0.1 * scene, -1.37, 2.5e-3], dtype=np.float32)
        self.voxel_size = 0.04 if scene % 2 else 0.02


class _FakeFuser:
    def __init__(self, scene):
        self.tsdf_fuser_pred = type("F", (), {"tsdf": _FakeTsdf(scene)})()


def _scene_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    ran, got = [], {}
    mine = par.run_scene_sharded(SCENE_FRAMES, lambda s: (ran.append(s), _FakeFuser(s))[1], world, rank,
                                 on_scene_done=lambda s, d: got.__setitem__(s, d), device="cpu")
    assert ran == mine
    torch.save({"mine": mine, "got": got}, os.path.join(out_dir, f"scenes{rank}.pt"))
    dist.destroy_process_group()


def test_scene_shard_and_tsdf_gather(tmp_path):
    world = 2
    plan = par.shard_scenes(SCENE_FRAMES, world)
    assert sorted(sum(plan, [])) == list(range(5)) and plan[0][0] == 0 and plan[1][0] == 2
    loads = [sum(SCENE_FRAMES[i] for i in p) for p in plan]
    assert max(loads) <= 52                                   # 92 frames: LPT gives 45/47 or similar, never worse than 4/3 OPT
    _spawn(_scene_worker, world, str(tmp_path))
    res = [torch.load(os.path.join(tmp_path, f"scenes{r}.pt"), weights_only=False) for r in range(world)]
    assert [r["mine"] for r in res] == plan
    assert res[1]["got"] == {}                                # only rank 0 receives the volumes
    got = res[0]["got"]
    assert sorted(got) == list(range(5))
    for s in range(5):
        want = _FakeTsdf(s)
        assert torch.equal(got[s]["tsdf_values"], want.tsdf_values) and torch.equal(got[s]["tsdf_weights"], want.tsdf_weights)
        np.testing.assert_array_equal(got[s]["origin_f32"], want.origin_f32)  # fp32 placement survives bit for bit
        assert got[s]["voxel_size"] == float(np.float32(want.voxel_size))


def test_gather_variable_single_process():
    p = torch.arange(5, dtype=torch.uint8)
    out = par.gather_variable(p, 1)
    assert len(out) == 1 and torch.equal(out[0], p)
    buf = par.pack_tsdf(torch.ones(8, 8, 8).half(), torch.zeros(8, 8, 8).half(), [1.5, -2.25, 3.0], 0.04, 7)
    d = par.unpack_tsdf(buf)
    assert d["scene_index"] == 7 and d["tsdf_values"].shape == (8, 8, 8) and float(d["tsdf_values"].sum()) == 512.0


# ---- keyframes in flight (parallel.KeyframePipeline) on two ranks -------------------------------------------------------
def _pipeline_worker(rank, world, port, out_dir):
    """run_two_pass(..., in_flight=4) on two ranks.  On the CPU the lanes are bookkeeping (no streams), which
    is exactly the part that must be rank independent: which lane a step gets, in which order the collectives are issued, in
    which order the gathered frames are integrated.  The spy records every collective; the replicas are integrated with the
    numpy TSDF oracle so that their bits can be compared with a serial, single-rank, in_flight=1 run."""
    _init(rank, world, port)
    calls = []
    real = dist.all_gather_into_tensor

    def spy(out, inp, *a, **kw):
        calls.append((tuple(out.shape), tuple(inp.shape), str(inp.dtype)))
        return real(out, inp, *a, **kw)

    par.dist.all_gather_into_tensor = spy
    hint, final = _OracleFuser(0.08), _OracleFuser(0.04)
    sf_hint = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=hint)
    sf_final = par.KeyframeShardFuser(torch.device("cpu"), world, rank, (H, W), fuser=final)
    lanes = []
    pipe = par.KeyframePipeline(torch.device("cpu"), in_flight=4, shard_fuser=sf_hint)
    step0 = pipe.step

    def step_spy(i, fn, **kw):
        lanes.append((i, pipe.lane_of(i)))
        return step0(i, fn, **kw)

    pipe.step = step_spy
    second = lambda i, state: tuple(t * 1.02 if j == 0 else t for j, t in enumerate(_batch(i)))
    n1, n2 = par.run_two_pass(NB, lambda i: SIZES[i], _batch, second, sf_hint, sf_final, between_passes=lambda f: "mesh",
                              pipeline=pipe)
    pipe.close()
    par.dist.all_gather_into_tensor = real
    assert (n1, n2) == (sum(SIZES), sum(SIZES))
    arrays = lambda f: [f.vol.values.copy(), f.vol.weights.copy()]
    torch.save({"calls": calls, "lanes": lanes, "hint": arrays(hint), "final": arrays(final)}, os.path.join(out_dir, f"pl{rank}.pt"))
    dist.destroy_process_group()


def test_two_pass_with_four_keyframes_in_flight_on_two_ranks(tmp_path):
    world = 2
    _spawn(_pipeline_worker, world, str(tmp_path))
    r0, r1 = (torch.load(os.path.join(tmp_path, f"pl{r}.pt"), weights_only=False) for r in range(world))
    steps = (NB + world - 1) // world
    # the same collectives, in the same order, on both ranks: one per step and pass
    assert r0["calls"] == r1["calls"] and len(r0["calls"]) == 2 * steps
    # lanes are dealt round-robin by STEP index -- identical on both ranks, restarting with every pass
    assert r0["lanes"] == r1["lanes"] == [(s, s % 4) for s in range(steps)] * 2
    # serial reference: one rank, one keyframe at a time
    hint, final = _OracleFuser(0.08), _OracleFuser(0.04)
    # (a single rank exchanges nothing, so nothing casts: apply the casts of OurFuser.fuse_frames / pack_update here)
    half = lambda f: (lambda d, K, T: f.fuse_frames(d.half(), K.half(), T.half()))
    sh = par.KeyframeShardFuser(torch.device("cpu"), 1, 0, (H, W), fuser=hint, fuse_fn=half(hint))
    sfin = par.KeyframeShardFuser(torch.device("cpu"), 1, 0, (H, W), fuser=final, fuse_fn=half(final))
    second = lambda i, state: tuple(t * 1.02 if j == 0 else t for j, t in enumerate(_batch(i)))
    par.run_two_pass(NB, lambda i: SIZES[i], _batch, second, sh, sfin, between_passes=lambda f: "mesh", in_flight=1)
    assert hint.vol.weights.sum() > 0 and final.vol.weights.sum() > 0
    for key, ref in (("hint", [hint.vol.values, hint.vol.weights]), ("final", [final.vol.values, final.vol.weights])):
        for a0, a1, want in zip(r0[key], r1[key], ref):
            np.testing.assert_array_equal(np.asarray(a0).view(np.uint16), np.asarray(want).view(np.uint16))
            np.testing.assert_array_equal(np.asarray(a1).view(np.uint16), np.asarray(want).view(np.uint16))


def _scene_inflight_worker(rank, world, port, out_dir):
    """run_scene_sharded(..., scans_in_flight=2, make_scan_fn=...): two scans per rank and round through
    loops.run_incremental_scans (CPU: lanes are bookkeeping), two gathers per round on every rank."""
    _init(rank, world, port)

    class FakeScan:  # the part of loops.IncrementalScan that run_incremental_scans / run_scene_sharded use
        def __init__(self, scene):
            self.scene, self.frames, self.fuser, self.left = scene, 0, _FakeFuser(scene), SCENE_FRAMES[scene]

        def step(self):
            if self.left == 0:
                return False
            order.append(self.scene)
            self.left -= 1
            self.frames += 1
            return True

    got, order = {}, []
    mine = par.run_scene_sharded(SCENE_FRAMES, None, world, rank, on_scene_done=lambda s, d: got.__setitem__(s, d), device="cpu",
                                 scans_in_flight=2, make_scan_fn=FakeScan)
    torch.save({"mine": mine, "got": got, "order": order}, os.path.join(out_dir, f"scenes_if{rank}.pt"))
    dist.destroy_process_group()


def test_scene_shard_with_two_scans_in_flight(tmp_path):
    world = 2
    _spawn(_scene_inflight_worker, world, str(tmp_path))
    res = [torch.load(os.path.join(tmp_path, f"scenes_if{r}.pt"), weights_only=False) for r in range(world)]
    plan = par.shard_scenes(SCENE_FRAMES, world)
    assert [r["mine"] for r in res] == plan
    # every scan ran to its end, two of a rank's scans at a time, frames of a pair interleaved round-robin
    for r in range(world):
        assert sorted(res[r]["order"]) == sorted(s for s in plan[r] for _ in range(SCENE_FRAMES[s]))
        first_pair = plan[r][:2]
        if len(first_pair) == 2:
            head = res[r]["order"][:2 * min(SCENE_FRAMES[s] for s in first_pair)]
            assert head == first_pair * (len(head) // 2)
    # rank 0 received every scan's TSDF, bit for bit
    assert sorted(res[0]["got"]) == list(range(len(SCENE_FRAMES))) and not res[1]["got"]
    for s, d in res[0]["got"].items():
        ref = _FakeTsdf(s)
        assert torch.equal(d["tsdf_values"], ref.tsdf_values) and torch.equal(d["tsdf_weights"], ref.tsdf_weights)


def _scene_batched_worker(rank, world, port, out_dir):
    """run_scene_sharded(..., scans_in_flight=2, make_scan_fn=..., batched_model_fn=...): the two scans of a round advance in
    lock step through loops.IncrementalScanBatch, one model call per turn on their collated frames."""
    _init(rank, world, port)

    class FakeScan:  # the part of loops.IncrementalScan that IncrementalScanBatch uses
        lookahead = None

        def __init__(self, scene):
            self.scene, self.frames, self.fuser, self.left, self.outs = scene, 0, _FakeFuser(scene), SCENE_FRAMES[scene], []

        def begin_frame(self):
            if self.left == 0:
                return None
            self.left -= 1
            return {"scene": torch.tensor([float(self.scene)]), "scan_id_string": f"s{self.scene}"}, {}

        def finish_frame(self, cur, out):
            assert out["depth"].shape == (1,) and float(out["depth"]) == 2.0 * self.scene
            self.frames += 1

    got, calls = {}, []

    def model_fn(cur, src):
        calls.append(list(cur["scan_id_string"]))
        return {"depth": 2.0 * cur["scene"]}

    mine = par.run_scene_sharded(SCENE_FRAMES, None, world, rank, on_scene_done=lambda s, d: got.__setitem__(s, d), device="cpu",
                                 scans_in_flight=2, make_scan_fn=FakeScan, batched_model_fn=model_fn)
    torch.save({"mine": mine, "got": got, "calls": calls}, os.path.join(out_dir, f"scenes_b{rank}.pt"))
    dist.destroy_process_group()


def test_scene_shard_with_two_scans_batched_per_model_call(tmp_path):
    world = 2
    _spawn(_scene_batched_worker, world, str(tmp_path))
    res = [torch.load(os.path.join(tmp_path, f"scenes_b{r}.pt"), weights_only=False) for r in range(world)]
    plan = par.shard_scenes(SCENE_FRAMES, world)
    assert [r["mine"] for r in res] == plan
    for r in range(world):
        want = []
        for i in range(0, len(plan[r]), 2):
            pair = plan[r][i:i + 2]
            for turn in range(max(SCENE_FRAMES[s] for s in pair)):  # the batch shrinks when the shorter scan of a pair ends
                want.append([f"s{s}" for s in pair if SCENE_FRAMES[s] > turn])
        assert res[r]["calls"] == want
    assert sorted(res[0]["got"]) == list(range(len(SCENE_FRAMES))) and not res[1]["got"]
    for s, d in res[0]["got"].items():
        ref = _FakeTsdf(s)
        assert torch.equal(d["tsdf_values"], ref.tsdf_values) and torch.equal(d["tsdf_weights"], ref.tsdf_weights)
