"""Multi-GPU partitioning of the hot path (no reference counterpart -- the
reference has no multi-GPU inference).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Three modes:

* **keyframe-batch shard** (throughput bench, offline two-pass -- 8e rows 1 and 3).  Keyframe batches
  are dealt round-robin (rank r takes batches r, r+W, ...).  Model evaluations are independent; only the
  TSDF accumulation couples them.  After each step every rank all-gathers the compact form of its TSDF
  update -- the predicted depth map plus K and cam_T_world, packed into ONE fp16 buffer -- and integrates
  the gathered frames of ALL ranks into its replica TSDF in canonical order (step-major, then rank =
  the serial batch order).  The order is fixed because the fp16 running mean and the weight clamp make
  integration order dependent (reference tools/tsdf.py:553-558), so every replica stays bit-identical
  to a single-GPU run over the same batch sequence.  ``run_sharded_pass`` / ``run_two_pass`` are the
  loops of reference test_offline_two_pass.py:26-131 (first pass) and :292-500 (second pass) in this form.
* **scene shard** (incremental mode -- 8e row 2).  Inside a scan frame t needs TSDF(t-1)
  (reference test_incremental.py:187-252,358-372), so scans -- not frames -- are dealt to ranks
  (``shard_scenes``: longest-processing-time by frame count).  No data-path collective; each finished
  scan's TSDF (values + weights, fp16, variable extent) goes to rank 0 with a size-then-padded
  all_gather (``gather_variable``) for export, in rounds so that the collectives match.
* **voxel-slab fusion** (``KeyframeShardFuser(mode="slab")`` -- 8e row 3, the alternative for large final volumes).
  The exchange is the same, but every rank integrates the gathered frames of ALL ranks into ITS x-slab of the volume
  only (X/W planes of Y*Z voxels: contiguous in the [X,Y,Z] layout); one all_gather of the slabs (``gather_slabs``)
  completes every replica before anything reads the volume (meshing, hint sampling, save_tsdf).  A voxel's update
  depends on that voxel and the frame sequence alone, and voxel centres are computed from the GLOBAL index
  (dt_tsdf_integrate_frames_xslab_f16), so the assembled volume is bit-identical to a serial run.  Replica fusion costs
  every rank N*b full-volume passes per step -- the one term of the design that grows with the GPU count (16 x 205 MB at
  cfg5's 0.02 m final volume on 8 GPUs); slab fusion keeps it at N*b/N passes' worth of voxels, for one 2*X*Y*Z*(2+2)/N
  byte gather per rank at the end of the pass.
* world == 1: nothing is exchanged and no packing kernel runs.
* **keyframes in flight** (``KeyframePipeline``): on every rank the keyframe batches of a pass run on several HIP streams
  (their latency-bound conv stacks overlap) while exchange + integration stay in batch order through an event chain;
  ``run_sharded_pass(..., in_flight=4)`` / ``run_two_pass(..., in_flight=4)``.  bench.py's timed loop is a client of it.

Payload per rank and step of the keyframe shard: b*(h*w + 32) halves (ScanNet depth-res 240x320:
154 KB per frame) -> latency-bound; a direct all_gather is one hop on the fully connected xGMI mesh.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# keyframe-batch shard
# ---------------------------------------------------------------------------------------------------
def pack_update(depth_b1hw: torch.Tensor, K_b44: torch.Tensor, cam_T_world_b44: torch.Tensor) -> torch.Tensor:
    """[b,1,h,w] depth + [b,4,4] K + [b,4,4] T -> [b, h*w + 32] fp16 (the casts OurFuser.fuse_frames applies)."""
    b = depth_b1hw.shape[0]
    return torch.cat([depth_b1hw.reshape(b, -1).half(), K_b44.reshape(b, 16).half(), cam_T_world_b44.reshape(b, 16).half()], 1)


def unpack_update(buf: torch.Tensor, h: int, w: int):
    n = buf.shape[0]
    depth = buf[:, : h * w].reshape(n, 1, h, w)
    K = buf[:, h * w: h * w + 16].reshape(n, 4, 4)
    T = buf[:, h * w + 16: h * w + 32].reshape(n, 4, 4)
    return depth, K, T


def shard_keyframes(num_frames: int, world: int, rank: int):
    """Round-robin keyframe-batch sharding: rank r takes frames r, r+world, ... (section 8e row 3)."""
    return list(range(rank, num_frames, world))


def _collective_ready():
    return dist.is_available() and dist.is_initialized()


def exchange_updates(local: torch.Tensor, world: int, force_collective: bool = False) -> torch.Tensor:
    """all_gather of the packed updates -> [world * b, n] in rank-major (canonical) order.
    force_collective: issue the collective even for world == 1 (single-GPU check of the RCCL path)."""
    if world == 1 and not (force_collective and _collective_ready()):
        return local
    out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())  # concatenation along dim 0 = rank-major order
    return out


def slab_bounds(X: int, world: int, rank: int):
    """x-range [x0, x1) of rank's slab: ceil(X / world) planes per rank, the last ranks possibly short or empty."""
    rows = (int(X) + world - 1) // world
    return min(rank * rows, int(X)), min((rank + 1) * rows, int(X))


def _slab_adapter(fuser):
    """(set_slab(x0, x1), arrays() -> tensors whose first dimension is X) of a fuser.  The HIP ``OurFuser`` restricts its
    integrate kernel through ``TSDFFuser.x_range``; test doubles bring their own ``set_slab`` / ``slab_arrays``."""
    if hasattr(fuser, "set_slab") and hasattr(fuser, "slab_arrays"):
        return fuser.set_slab, fuser.slab_arrays
    tf = fuser.tsdf_fuser_pred
    X = int(tf.tsdf.tsdf_values.shape[0])

    def set_slab(x0, x1):
        tf.x_range = (int(x0), int(x1))

    def arrays():
        t = tf.tsdf
        return [t.tsdf_values, t.tsdf_weights, t.voxel_bitmap.view(X, -1)]  # (Y*Z is a multiple of 64: whole words per plane)

    return set_slab, arrays


class KeyframeShardFuser:
    """Replica TSDF + per-step exchange (``mode="replica"``), or voxel-slab fusion (``mode="slab"``: see the module text;
    needs ``fuser`` -- the HIP ``OurFuser`` or an object with ``set_slab(x0, x1)`` / ``slab_arrays()``; call
    ``gather_slabs()`` before the volume is read).

    ``fuse_fn(depth_n1hw, K_n44, T_n44)`` integrates n frames in the given order; by default it is
    ``fuser.fuse_frames`` of the HIP ``OurFuser`` passed as ``fuser``.  ``depth_hw`` is the size of the depth maps
    that travel (every rank must use the same); ``upsample_to=(H, W)`` applies the drivers' nearest-neighbour
    upsampling to the ground-truth depth size (reference test_offline_two_pass.py:97-101) AFTER the exchange, so
    only the predicted resolution crosses xGMI.
    """

    def __init__(self, device, world, rank, depth_hw, fuser=None, fuse_fn=None, force_collective=False, upsample_to=None,
                 mode="replica"):
        if mode not in ("replica", "slab"):
            raise ValueError(f"mode {mode!r}: 'replica' or 'slab'")
        self.mode = mode
        self.world, self.rank, self.device = int(world), int(rank), device
        self.force_collective = bool(force_collective)
        self.h, self.w = int(depth_hw[0]), int(depth_hw[1])
        self.upsample_to = None if upsample_to is None else (int(upsample_to[0]), int(upsample_to[1]))
        self.fuser = fuser
        if fuse_fn is None:
            if fuser is None:
                raise ValueError("KeyframeShardFuser needs a fuser or a fuse_fn")
            fuse_fn = lambda d, K, T: fuser.fuse_frames(d, K, T, None)
        self.fuse_fn = fuse_fn
        self._local = None
        self._all = None
        self.frames_fused = 0
        self.slabs_current = True  # (slab mode: False between the first integration and the next gather_slabs())
        if mode == "slab":
            if fuser is None:
                raise ValueError("slab mode needs the fuser whose volume is sharded")
            self._set_slab, self._slab_arrays = _slab_adapter(fuser)
            self._X = int(self._slab_arrays()[0].shape[0])
            self.slab = slab_bounds(self._X, self.world, self.rank)
            self._set_slab(*self.slab)

    def _fuse(self, depth, K, T):
        if self.upsample_to is not None and tuple(depth.shape[-2:]) != self.upsample_to:
            depth = F.interpolate(depth.float(), size=self.upsample_to, mode="nearest").to(depth.dtype)
        self.fuse_fn(depth, K, T)
        self.frames_fused += int(depth.shape[0])
        if self.mode == "slab":
            self.slabs_current = False

    def gather_slabs(self):
        """Slab mode: all_gather every rank's x-slab (values, weights, active bits) so that each replica holds the whole
        volume again.  Collective: every rank must call it at the same point (the loops below do, at the end of a pass).
        Integration may continue afterwards (the replica outside the own slab goes stale again until the next gather).
        Returns the number of bytes this rank received."""
        if self.mode != "slab":
            return 0
        collective = self.world > 1 or (self.force_collective and _collective_ready())
        got = 0
        if collective:
            rows = (self._X + self.world - 1) // self.world
            x0, x1 = self.slab
            for a in self._slab_arrays():
                send = torch.zeros((rows,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
                send[: x1 - x0].copy_(a[x0:x1])
                out = torch.empty((self.world * rows,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
                # (as raw bytes: the payload is bit patterns -- half values, int32 bitmap words -- and byte tensors are the
                #  one dtype every backend moves)
                dist.all_gather_into_tensor(out.view(torch.uint8), send.view(torch.uint8))
                a.copy_(out[: self._X])  # rank-major rows == x order; rows past X are the padding of a short last slab
                got += out.numel() * out.element_size()
        self.slabs_current = True
        return got

    def exchange_and_fuse(self, depth_b1hw, K_b44, cam_T_world_b44, counts=None, rows=None):
        """One step.  ``depth_b1hw`` [b,1,h,w] with its cameras ([b,4,4] each, any float dtype; cast to half like
        OurFuser.fuse_frames), or None when this rank has no batch in this step.  ``counts``: the number of frames
        every rank contributes in this step (length world; known from the schedule, so no flag travels); default =
        every rank contributes b.  ``rows``: rows of the exchange buffer (max batch size), default max(counts).
        Returns the number of frames integrated."""
        b = 0 if depth_b1hw is None else int(depth_b1hw.shape[0])
        if counts is None:
            counts = [b] * self.world
        if len(counts) != self.world or counts[self.rank] != b:
            raise ValueError(f"counts {counts} does not match world {self.world} / this rank's batch of {b}")
        if b and tuple(depth_b1hw.shape[-2:]) != (self.h, self.w):
            raise ValueError(f"depth maps are {tuple(depth_b1hw.shape[-2:])}, the exchange was set up for {(self.h, self.w)}")
        collective = self.world > 1 or (self.force_collective and _collective_ready())
        if not collective:
            if b:  # single GPU: nothing to exchange -- integrate the own frames directly (no packing kernels)
                self._fuse(depth_b1hw, K_b44, cam_T_world_b44)
            return b
        rows = max(counts) if rows is None else int(rows)
        if rows == 0:
            return 0
        n = self.h * self.w
        if self._local is None or self._local.shape[0] != rows:
            self._local = torch.zeros((rows, n + 32), dtype=torch.float16, device=self.device)
            self._all = torch.empty((self.world * rows, n + 32), dtype=torch.float16, device=self.device)
        if b:  # converting copies straight into the send buffer: depth, then [K | T]
            self._local[:b, :n].copy_(depth_b1hw.reshape(b, n))
            self._local[:b, n:n + 16].copy_(K_b44.reshape(b, 16))
            self._local[:b, n + 16:].copy_(cam_T_world_b44.reshape(b, 16))
        dist.all_gather_into_tensor(self._all, self._local)  # rank-major = canonical batch order
        if all(c == rows for c in counts):
            sel = self._all
        else:  # ragged tail of the schedule: drop the padding rows (host-known indices, no sync)
            idx = [r * rows + i for r, c in enumerate(counts) for i in range(c)]
            sel = self._all[torch.as_tensor(idx, device=self._all.device)]
        depth, K, T = unpack_update(sel, self.h, self.w)
        self._fuse(depth, K, T)
        return int(depth.shape[0])


#: keyframe batches in flight per GPU for batch-1 workloads (round 5: 742 frames/s at 4 against 700 at 2 and 730 at 3 at
#: 640x480, profiles/r4z_streams_probe.txt); a batch of 8 already fills the chip -- use 1 there
DEFAULT_IN_FLIGHT = 4
#: in_flight -> how many steps the host may run ahead of the GPU (KeyframePipeline max_lead="auto"); others: in_flight + 1
DEFAULT_MAX_LEAD = {1: 2}


class KeyframePipeline:
    """Several independent keyframe batches in flight on one GPU -- the schedule of the headline number as a product feature.

    Keyframe batches of the offline passes (reference test_offline_two_pass.py:76-126 and :300-470) do not depend on each
    other: hints and cameras are inputs, only the TSDF accumulation is order dependent.  At batch 1 a keyframe is a chain of
    ~50 latency-bound kernels that leaves most of the chip idle, so the pipeline runs keyframe i on HIP stream
    ``i % in_flight`` ("lane") and the conv stacks of neighbouring keyframes overlap (602 -> 760 frames/s at 640x480 with 4
    lanes).  The pipeline owns

      * the lane streams (created once; allocator pools and per-stream library scratch warm up on their first steps),
      * the in-order fuse chain: the exchange + TSDF integration of keyframe i is enqueued on ITS lane behind an event that
        closes keyframe i-1's integration, so the replica sees the frames in the serial batch order (the fp16 running mean
        and the weight clamp make integration order dependent, reference tools/tsdf.py:553-558) and -- with several ranks --
        the per-step collectives are issued and executed in one order on every rank, one at a time,
      * the conv plan objective for its lifetime (``conv_ops.PLAN_THROUGHPUT`` while more than one keyframe is in flight:
        launch plans that leave room for the other lanes' workgroups; restored by ``close()``),
      * the hardware-queue requirement: the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues,
        so a fifth stream (4 lanes + the default one) shares a queue with a lane (647 instead of 742 frames/s).  The variable
        is read when the runtime initialises: call ``doubletake_amd.hwqueues.ensure(in_flight)`` before importing torch; the
        constructor warns if it finds fewer queues than lanes + 1.

    Cross-lane data (matching features of the HBM feature cache, packed weights) orders itself through the ready events its
    producers attach (``_abi.record_ready`` / ``wait_ready``); nothing here synchronises the host.

    ``model``: optional hot-path model; ``launch_programs=True`` switches it to recorded launch programs
    (``enable_launch_programs``: one C call per model step instead of ~50 launches from Python; each lane records its own
    program and owns its output buffers, which the lane's fuse consumes before the lane's next step overwrites them).

    On a CPU device (gloo tests of the multi-rank logic) lanes are bookkeeping only: steps run in submission order.

    Use as a context manager, or call ``close()``.  ``step(i, fn)`` runs ``fn()`` -> (depth_b1hw, K_b44, cam_T_world_b44) (or
    None: this rank has no batch in this step) on lane i and enqueues the exchange + integration behind it; ``drain()`` makes
    the caller's stream wait for every lane (end of a pass: before meshing, hint sampling, saving)."""

    def __init__(self, device, in_flight=DEFAULT_IN_FLIGHT, shard_fuser=None, conv_plan="auto", model=None,
                 launch_programs=False, max_lead="auto"):
        self.device = torch.device(device)
        self.in_flight = max(1, int(in_flight))
        # back-pressure: the host may be at most `max_lead` submitted-but-unfinished steps ahead of the GPU (it blocks on the
        # completion event of step i - max_lead before submitting step i).  Bounds the queued work (and the latency of
        # everything queued behind it), and -- measured, profiles/r6f_lead_probe.txt -- the GPU itself is faster when a lane's
        # next step is not already waiting behind the running one: with the model step enqueued by one C call the host
        # otherwise runs dozens of steps ahead.  "auto": DT_PIPE_LEAD or the default below; None / 0: unbounded.
        if max_lead == "auto":
            env = os.environ.get("DT_PIPE_LEAD")
            max_lead = int(env) if env else DEFAULT_MAX_LEAD.get(self.in_flight, self.in_flight + 1)
        self.max_lead = int(max_lead) if max_lead else 0
        self._pending = []
        self.host_wait_s = 0.0
        self.shard_fuser = shard_fuser
        self.cuda = self.device.type == "cuda"
        self.streams = None
        self._fuse_done = None
        self._steps = 0
        self._last_lane = 0
        self._prev_plan = None
        self._closed = False
        self.model = model
        if self.cuda and self.in_flight > 1:
            caller = torch.cuda.current_stream(self.device)
            self.streams = [torch.cuda.Stream(self.device) for _ in range(self.in_flight)]
            for st in self.streams:
                st.wait_stream(caller)
            from . import hwqueues

            hwqueues.check(self.in_flight)
        if self.cuda and conv_plan is not None:
            from .modules import conv_ops

            if conv_plan == "auto":
                mask = conv_ops.PLAN_THROUGHPUT if self.in_flight > 1 else conv_ops.PLAN_LATENCY
            elif conv_plan in ("latency", "throughput"):
                mask = conv_ops.PLAN_THROUGHPUT if conv_plan == "throughput" else conv_ops.PLAN_LATENCY
            else:
                mask = int(conv_plan)
            self._prev_plan = conv_ops.current_plan_objective()
            self.conv_plan_mask = conv_ops.set_plan_objective(mask)
        else:
            self.conv_plan_mask = None
        if model is not None and launch_programs and self.cuda:
            model.enable_launch_programs(True)
        # volume gate (experiment switch DT_PIPE_GATE=volume; default off): keyframe i+1's volume kernel waits for keyframe i's.
        # Built to stagger the lanes on the GPU once the host no longer does it by being slow; measured no better than the
        # hardware's own interleaving (profiles/r6d_gate_probe3.txt: 755 vs 761 frames/s at 3 lanes) -- what does help is
        # max_lead above
        self._vol_done = None
        self.gate = os.environ.get("DT_PIPE_GATE", "off") if (self.streams is not None and model is not None) else "off"
        if self.gate != "off" and hasattr(model, "cost_volume"):
            model.cost_volume.__dict__["_stage_hook"] = self._gate

    def _gate(self, tag):
        cur = torch.cuda.current_stream(self.device)
        if tag == "mlp_begin":
            if self._vol_done is not None:
                cur.wait_event(self._vol_done)
        elif tag == "mlp_end":
            ev = torch.cuda.Event()
            ev.record(cur)
            self._vol_done = ev

    # -- lanes -------------------------------------------------------------------------------------------------------
    def lane_of(self, i):
        return int(i) % self.in_flight

    def lane(self, i):
        """Context manager: torch's current stream is lane i's stream inside (a no-op with one lane or on the CPU)."""
        if self.streams is None:
            import contextlib

            return contextlib.nullcontext()
        return torch.cuda.stream(self.streams[self.lane_of(i)])

    def step(self, i, fn, counts=None, rows=None):
        """Keyframe batch i: ``fn()`` on lane i, then exchange + integrate in batch order.  Returns (fn's result, number of
        frames integrated)."""
        if self._closed:
            raise RuntimeError("KeyframePipeline.step after close()")
        if self.cuda and self.max_lead and len(self._pending) >= self.max_lead:
            t0 = time.perf_counter()
            self._pending.pop(0).synchronize()
            self.host_wait_s += time.perf_counter() - t0  # (blocked on the GPU, not issuing: bench.py subtracts it)
        self._last_lane = self.lane_of(i)
        with self.lane(i):
            res = fn()
            n = 0
            if self.shard_fuser is not None:
                depth, K, T = res if res is not None else (None, None, None)
                n = self.fuse(depth, K, T, counts=counts, rows=rows)
            if self.cuda and self.max_lead:
                done = self._fuse_done if (self.shard_fuser is not None and self.streams is not None) else None
                if done is None:
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                self._pending.append(done)
        self._steps += 1
        return res, n

    def fuse(self, depth, K, T, counts=None, rows=None):
        """Exchange + integrate on the CURRENT stream (a lane), ordered behind the previous keyframe's integration."""
        if self.streams is None:
            return self.shard_fuser.exchange_and_fuse(depth, K, T, counts=counts, rows=rows)
        cur = torch.cuda.current_stream(self.device)
        if self._fuse_done is not None:
            cur.wait_event(self._fuse_done)
        n = self.shard_fuser.exchange_and_fuse(depth, K, T, counts=counts, rows=rows)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._fuse_done = ev
        return n

    def drain(self):
        """The caller's current stream waits for everything enqueued on the lanes (no host synchronisation)."""
        if self.streams is None:
            return
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            if st != cur:
                cur.wait_stream(st)

    def finish_pass(self):
        """End of a pass: complete the replicas (slab mode: ONE gather of the x-slabs, issued on the lane that ran the last
        integration so that it follows it), then ``drain()``.  Collective in slab mode: every rank calls it."""
        if self.shard_fuser is not None:
            if self.streams is not None and self._steps:
                with torch.cuda.stream(self.streams[self._last_lane]):
                    if self._fuse_done is not None:
                        torch.cuda.current_stream(self.device).wait_event(self._fuse_done)
                    self.shard_fuser.gather_slabs()
            else:
                self.shard_fuser.gather_slabs()
        self.drain()

    def close(self):
        if self._closed:
            return
        self._closed = True
        self.drain()
        if self.gate != "off" and self.model is not None and self.model.cost_volume.__dict__.get("_stage_hook") == self._gate:
            del self.model.cost_volume.__dict__["_stage_hook"]
        if self._prev_plan is not None:
            from .modules import conv_ops

            conv_ops.set_plan_objective(self._prev_plan)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def run_sharded_pass(num_batches, batch_size_of, step_fn, shard_fuser: KeyframeShardFuser, in_flight=1, pipeline=None):
    """One pass over a scan's keyframe batches, sharded over the ranks (the loop of reference
    test_offline_two_pass.py:76-126 / :300-470 with the fuser call replaced by exchange + replica integrate).

    ``step_fn(batch_index) -> (depth_b1hw, K_b44, cam_T_world_b44)`` evaluates one batch (hint preparation + model);
    it only runs for this rank's batches.  ``batch_size_of(i)`` is the number of keyframes in batch i (all ranks know
    the dataloader's length and batch size; the last batch may be short).  ``in_flight``: keyframe batches in flight on
    this GPU (``KeyframePipeline``; 1 = strictly one after the other, ``DEFAULT_IN_FLIGHT`` for batch-1 passes); or pass a
    ``pipeline`` built by the caller (its shard fuser is replaced by ``shard_fuser`` for the pass).  Results are
    bit-identical whatever the value: only the order of integration matters, and it is kept.  Returns the number of frames
    integrated into the replica (= all frames of the scan, on every rank)."""
    world, rank = shard_fuser.world, shard_fuser.rank
    sizes = [int(batch_size_of(i)) for i in range(num_batches)]
    rows = max(sizes) if sizes else 0
    total = 0
    own = pipeline is None
    pipe = KeyframePipeline(shard_fuser.device, in_flight=in_flight, shard_fuser=shard_fuser) if own else pipeline
    prev_fuser, pipe.shard_fuser = pipe.shard_fuser, shard_fuser
    try:
        for s in range((num_batches + world - 1) // world):
            counts = [sizes[s * world + r] if s * world + r < num_batches else 0 for r in range(world)]
            mine = s * world + rank
            fn = (lambda i=mine: step_fn(i)) if mine < num_batches else (lambda: None)
            _, n = pipe.step(s, fn, counts=counts, rows=rows)
            total += n
        pipe.finish_pass()  # (slab mode: complete every replica before anything reads the volume; then drain the lanes)
    finally:
        pipe.shard_fuser = prev_fuser
        if own:
            pipe.close()
    return total


def run_two_pass(num_batches, batch_size_of, first_pass_fn, second_pass_fn, hint_shard_fuser, final_shard_fuser,
                 between_passes=None, num_first_batches=None, first_batch_size_of=None, in_flight=1, pipeline=None):
    """Offline two-pass over one scan (reference test_offline_two_pass.py:26-131 then :292-500).

    Pass 1 (``first_pass_fn``: model with empty hints) fills every rank's replica of the 0.04 m / 3 m hint TSDF
    (``hint_shard_fuser``; reference :48-53).  The replicas are bit-identical, so the hint mesh
    (``get_mesh_pytorch3d``, reference :129) is extracted locally on every rank -- no collective --
    in ``between_passes(hint_fuser)`` whose result is handed to ``second_pass_fn(batch_index, hint_state)``.
    Pass 2 renders hints from that mesh, samples the hint TSDF's weights, runs the model and fuses into the final
    volume (``final_shard_fuser``, may be None when fusion is off).

    ``in_flight`` / ``pipeline``: keyframe batches in flight per GPU in both passes (``KeyframePipeline``; the lanes are
    drained between the passes, before the hint mesh is extracted).

    Revisit flow (reference test_revisit.py:104-260, ``loops.revisit_fns``): the first pass runs over ANOTHER scan --
    ``num_first_batches`` / ``first_batch_size_of`` describe that scan's keyframe batches (default: the same schedule)."""
    own = pipeline is None and in_flight > 1
    pipe = KeyframePipeline(hint_shard_fuser.device, in_flight=in_flight, shard_fuser=hint_shard_fuser) if own else pipeline
    try:
        nb1 = num_batches if num_first_batches is None else num_first_batches
        n1 = run_sharded_pass(nb1, first_batch_size_of if first_batch_size_of is not None else batch_size_of, first_pass_fn,
                              hint_shard_fuser, pipeline=pipe)
        state = between_passes(hint_shard_fuser.fuser) if between_passes is not None else None
        if final_shard_fuser is None:
            mine = shard_keyframes(num_batches, hint_shard_fuser.world, hint_shard_fuser.rank)
            if pipe is None:
                for i in mine:
                    second_pass_fn(i, state)
            else:
                keep, pipe.shard_fuser = pipe.shard_fuser, None
                try:
                    for j, i in enumerate(mine):
                        pipe.step(j, lambda i=i: second_pass_fn(i, state))
                    pipe.drain()
                finally:
                    pipe.shard_fuser = keep
            return n1, 0
        n2 = run_sharded_pass(num_batches, batch_size_of, lambda i: second_pass_fn(i, state), final_shard_fuser, pipeline=pipe)
        return n1, n2
    finally:
        if own:
            pipe.close()


# ---------------------------------------------------------------------------------------------------
# scene shard (incremental mode)
# ---------------------------------------------------------------------------------------------------
def shard_scenes(frame_counts, world: int):
    """Deal scans to ranks, longest first onto the least loaded rank (ties -> lowest rank / lowest scan index):
    deterministic, and within 4/3 of the optimal makespan.  Returns ``world`` lists of scan indices, each in
    the order that rank processes them."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += int(frame_counts[i])
    return out


_HDR = 8  # int32 words: X, Y, Z, scene index, origin.x/y/z (fp32 bits), voxel size (fp32 bits)


def pack_tsdf(values: torch.Tensor, weights: torch.Tensor, origin_f32, voxel_size: float, scene_index: int) -> torch.Tensor:
    """fp16 values + weights [X,Y,Z] and the volume's placement -> one flat byte buffer (bit copies, nothing rounds;
    uint8 because it is an element type both RCCL and gloo move)."""
    X, Y, Z = values.shape
    hdr = np.zeros(_HDR, dtype=np.int32)
    hdr[:4] = (X, Y, Z, scene_index)
    hdr[4:7] = np.asarray(origin_f32, dtype=np.float32).view(np.int32)
    hdr[7] = np.float32(voxel_size).view(np.int32)
    h8 = torch.from_numpy(hdr.view(np.uint8).copy()).to(values.device)
    return torch.cat([h8, values.contiguous().half().reshape(-1).view(torch.uint8),
                      weights.contiguous().half().reshape(-1).view(torch.uint8)])


def unpack_tsdf(buf: torch.Tensor):
    hdr = buf[: 4 * _HDR].cpu().numpy().view(np.int32)
    X, Y, Z, scene = (int(v) for v in hdr[:4])
    n = 2 * X * Y * Z  # bytes per volume
    body = buf[4 * _HDR: 4 * _HDR + 2 * n]
    return dict(scene_index=scene, origin_f32=hdr[4:7].copy().view(np.float32), voxel_size=float(hdr[7:8].copy().view(np.float32)[0]),
                tsdf_values=body[:n].view(torch.float16).reshape(X, Y, Z), tsdf_weights=body[n:].view(torch.float16).reshape(X, Y, Z))


def _collective_device():
    """Device collectives of the default process group move: the current GPU under nccl (= RCCL), else the CPU."""
    if _collective_ready() and "nccl" in str(dist.get_backend()):
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_variable(payload: torch.Tensor, world: int, dst: int = 0, rank: int = 0):
    """Variable-size gather as two all_gathers: element counts first, then the payloads padded to the largest.
    ``payload`` is 1-D (may be empty).  Returns the list of ``world`` payloads on ``dst``, None elsewhere.
    (all_gather rather than gather: it is the collective the north star names, and on the xGMI mesh its cost is
    the same single hop; with 288 GB per GPU the padded receive buffer is not a concern.)"""
    if world == 1 and not _collective_ready():
        return [payload]
    dev = payload.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([payload.numel()], dtype=torch.int64, device=dev))
    sizes = [int(v) for v in sizes.tolist()]
    cap = max(sizes)
    if cap == 0:
        return [payload.new_empty(0) for _ in range(world)] if rank == dst else None
    send = payload.new_zeros(cap)
    send[: payload.numel()] = payload
    recv = payload.new_empty(world * cap)
    dist.all_gather_into_tensor(recv, send)
    if rank != dst:
        return None
    return [recv[r * cap: r * cap + sizes[r]] for r in range(world)]


def run_scene_sharded(frame_counts, run_scene_fn, world: int, rank: int, on_scene_done=None, device=None, scans_in_flight=1,
                      make_scan_fn=None, batched_model_fn=None):
    """Incremental mode over a scan list (reference test_incremental.py:114-491, the ``for scan in scans`` loop).

    ``run_scene_fn(scene_index)`` runs one scan's sequential per-frame loop (hint from TSDF(t-1) -> model -> fuse,
    reference :172-372) on this rank and returns its fuser (anything with ``tsdf_fuser_pred.tsdf``) or None.  After
    every round (the i-th scan of each rank) the finished TSDFs are gathered to rank 0, where
    ``on_scene_done(scene_index, dict(tsdf_values, tsdf_weights, origin_f32, voxel_size))`` receives them in ascending
    scan order of the round.  Returns this rank's list of scan indices.

    Round 6 -- scans in flight per GPU: with ``make_scan_fn(scene_index) -> loops.IncrementalScan`` and
    ``scans_in_flight = k > 1`` a rank takes k scans of its list per round and runs them interleaved on k HIP-stream lanes
    (``loops.run_incremental_scans``: scans are independent, one scan alone is a chain of latency-bound kernels; 721 -> 882
    frames/s per GPU at 512x384 with four, ``profiles/r6q_time_incremental_scans_program.json``), then the k finished TSDFs
    are gathered one after the other (every rank issues k gathers per round, empty payloads where it has no scan).
    With ``batched_model_fn`` the k scans of a round advance in lock step instead, their current frames evaluated by ONE
    ``batched_model_fn(cur_data_k, src_data_k)`` call per turn (``loops.IncrementalScanBatch``: 1003 frames/s with four scans,
    1107 with eight, ``profiles/r7b_*``; per-scan results equal the scan alone to fp32 rounding, not bit for bit)."""
    plan = shard_scenes(frame_counts, world)
    k = max(1, int(scans_in_flight)) if make_scan_fn is not None else 1
    rounds = max(((len(p) + k - 1) // k for p in plan), default=0)
    for i in range(rounds):
        mine = plan[rank][i * k:(i + 1) * k]
        fusers = []
        if mine:
            if make_scan_fn is not None:
                from . import loops

                scans = [make_scan_fn(scene) for scene in mine]
                if batched_model_fn is not None:
                    loops.run_incremental_scans([loops.IncrementalScanBatch(scans, batched_model_fn)], in_flight=1, device=device)
                else:
                    loops.run_incremental_scans(scans, in_flight=k, device=device)
                fusers = [(scene, sc.fuser) for scene, sc in zip(mine, scans)]
            else:
                fusers = [(scene, run_scene_fn(scene)) for scene in mine]
        for j in range(k):
            payload = None
            if j < len(fusers) and fusers[j][1] is not None:
                scene, fuser = fusers[j]
                t = fuser.tsdf_fuser_pred.tsdf
                payload = pack_tsdf(t.tsdf_values, t.tsdf_weights, t.origin_f32, t.voxel_size, scene)
            if payload is None:
                # a rank without a scan in this slot still takes part in the collective, on the device the backend moves
                # (RCCL: this rank's GPU; a CPU tensor here would error or hang the nccl all_gather -- ADVICE r2)
                payload = torch.empty(0, dtype=torch.uint8, device=device if device is not None else _collective_device())
            got = gather_variable(payload, world, dst=0, rank=rank)
            if rank == 0 and on_scene_done is not None:
                done = [unpack_tsdf(g) for g in got if g.numel()]
                for d in sorted(done, key=lambda d: d["scene_index"]):
                    on_scene_done(d.pop("scene_index"), d)
    return plan[rank]
