"""Per-scan evaluation loops around the hot path, with the data contracts of the reference drivers.

``run_incremental_scan``  -- reference test_incremental.py:172-372: frame t's hint is rendered from the TSDF fused
                            from frames < t (marching cubes -> depth render -> weight sampling), then model, then fuse.
``run_incremental_scans`` -- several such scans in flight on one GPU (``IncrementalScan`` objects on HIP-stream lanes):
                            scans are independent, the frames of one scan are not.
``two_pass_fns``          -- reference test_offline_two_pass.py:26-131 (first pass, empty hints, hint TSDF at
                            0.04 m / 3 m) and :292-500 (second pass, hints from the finished first-pass mesh) as the
                            two step functions ``parallel.run_two_pass`` shards over GPUs.
``revisit_fns``           -- reference test_revisit.py:104-260: the hint mesh comes from a first pass over a PREVIOUS scan
                            of the same place (:122-156, compute_hint_mesh), the second pass runs over the new scan with
                            hints rendered from that mesh through the rigid transform between the two scans' world frames.
``FrameTimer``            -- the per-frame ``model_time`` / ``hint_time`` the reference drivers record with CUDA events
                            (test_incremental.py:107-111,274-288; test_revisit.py:233-256) and average into the score sheet.

The dataset / dataloader, metric averaging, visualisation and file output of those scripts are out of scope
(SURVEY.md section 2); a batch here is the pair of dicts ``(cur_data, src_data)`` the reference dataloaders yield,
already on the GPU.  ``model_fn(cur_data, src_data) -> outputs`` is ``lambda c, s: model("test", c, s,
return_mask=True)`` for a DepthModelCVHint with an image encoder attached, or any callable with that contract.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _abi
from .tools import tsdf as _tsdf_mod
from .utils.rendering_utils import MeshDepthRenderer, empty_hint, prepare_mesh_hint, prepare_mesh_hint_fused


class FrameTimer:
    """``model_time`` / ``hint_time`` per frame in milliseconds, from HIP events on the current stream.

    The reference brackets the hint preparation and the model call with CUDA events and calls
    ``torch.cuda.synchronize()`` after each to read them (test_incremental.py:107-111,205,256-258,274-288), then feeds
    ``elapsed / batch_size`` per element into its ResultsAverager.  Here the events are kept and read once, in
    ``summary()`` -- the loop itself stays free of host synchronisation (``sync_each_frame=True`` restores the
    reference's behaviour for like-for-like timing).  ``write_scores`` stores the averages through the reference's score
    sheet format (utils/formats.py:write_scores_json = utils/metrics_utils.py ResultsAverager.output_json)."""

    def __init__(self, sync_each_frame=False):
        self.sync_each_frame = sync_each_frame
        self._open = {}
        self._spans = {"hint_time": [], "model_time": []}

    def start(self, what):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._open[what] = ev

    def stop(self, what, batch_size=1):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._spans[what].append((self._open.pop(what), ev, int(batch_size)))
        if self.sync_each_frame:
            torch.cuda.synchronize()

    def skip(self, what, batch_size=1):
        """A frame without this phase (the first frame of the incremental mode has no hint to render): 0 ms."""
        self._spans[what].append((None, None, int(batch_size)))

    def per_frame(self):
        """{"hint_time": [...], "model_time": [...]}: one entry per batch ELEMENT (elapsed / batch size, as the reference)."""
        torch.cuda.synchronize()
        out = {}
        for what, spans in self._spans.items():
            vals = []
            for a, b, n in spans:
                ms = 0.0 if a is None else a.elapsed_time(b)
                vals += [ms / n] * n
            out[what] = vals
        return out

    def summary(self):
        pf = self.per_frame()
        return {k: (sum(v) / len(v) if v else 0.0) for k, v in pf.items()}

    def write_scores(self, filepath, exp_name, metrics_name="frame metrics", extra=None):
        from .utils import formats

        scores = dict(extra or {})
        scores.update(self.summary())
        return formats.write_scores_json(filepath, exp_name, metrics_name, scores)


def _depth_for_fusion(outputs, size, mask_pred_depth=False, per_view_mask=True):
    """Nearest upsampling of depth_pred_s0 to the ground-truth depth size and the optional masking
    (test_incremental.py:290-294,331-352; test_offline_two_pass.py:97-117)."""
    depth = outputs["depth_pred_s0_b1hw"]
    if size is not None and tuple(depth.shape[-2:]) != tuple(size):
        depth = F.interpolate(depth, size=tuple(size), mode="nearest")
    if mask_pred_depth:
        m = outputs["overall_mask_bhw"].float()
        if per_view_mask and m.dim() == 4:  # slow manager: per-view masks [b,K,h,w]; valid where more than two views agree
            m = F.interpolate(m, size=depth.shape[-2:], mode="nearest").bool().sum(1, keepdim=True) > 2
        else:
            m = F.interpolate(m.view(m.shape[0], 1, *m.shape[-2:]), size=depth.shape[-2:], mode="nearest").bool()
        depth = depth.clone()
        depth[~m] = -1
    return depth


@torch.no_grad()
def matching_lookahead(model):
    """``lookahead`` callable for run_incremental_scan: encodes the NEXT frame's new keyframe image into the model's HBM
    feature cache on a side stream (DepthModelCVHint.prefetch_matching_feats) while the current frame is still running.
    Needs ``model.use_feature_cache = True`` and batches that carry ``frame_id_string`` (pass_frame_id=True)."""
    model.use_feature_cache = True

    def run(cur_data, src_data):
        model.prefetch_matching_feats(cur_data["image_b3hw"], list(cur_data["frame_id_string"]),
                                      scan_ids=cur_data.get("scan_id_string"))

    # the loop hands the call to the model, which issues it right behind the current frame's volume kernel: the encoder
    # then runs beside the conv stack instead of delaying the volume kernel (which needs every CU's LDS to itself)
    run.after_volume_of = model
    return run


class IncrementalScan:
    """One scan of the incremental (online) mode as a step-able object: ``step()`` processes the next frame on torch's CURRENT
    stream (hint from the TSDF fused so far -> model -> fuse) and returns False once the scan is exhausted.
    ``run_incremental_scan`` drives one of these to the end; ``run_incremental_scans`` keeps several scans in flight on HIP
    streams (different scans do not depend on each other; the frames of one scan do).  Arguments: see run_incremental_scan."""

    def __init__(self, model_fn, fuser, batches, render_hw, fuse_size=None, fused_hint=True, mask_pred_depth=False,
                 on_frame=None, timer: FrameTimer | None = None, lookahead=None):
        self.model_fn, self.fuser = model_fn, fuser
        self.H2, self.W2 = render_hw
        self.fuse_size, self.fused_hint, self.mask_pred_depth = fuse_size, fused_hint, mask_pred_depth
        self.on_frame, self.timer, self.lookahead = on_frame, timer, lookahead
        self.renderer = None if fused_hint else MeshDepthRenderer(self.H2, self.W2)
        self.frames = 0
        self._it = iter(batches)
        self._next = self._pull() if lookahead is not None else None  # (a lookahead needs the batch after the current one)

    def _pull(self):
        try:
            return next(self._it)
        except StopIteration:
            return None

    @torch.no_grad()
    def begin_frame(self, item=None):
        """Pull the next frame and prepare its hint (on torch's current stream): returns (cur_data, src_data), or None when the
        scan is exhausted.  Followed by the model call and ``finish_frame(cur_data, outputs)`` -- ``step()`` does all three;
        ``run_incremental_scans_batched`` puts the frames of several scans through ONE model call in between."""
        if item is None:
            item = self._pull()
        if item is None:
            return None
        cur_data, src_data = item
        fuser, timer = self.fuser, self.timer
        H2, W2 = self.H2, self.W2
        if cur_data["cam_T_world_b44"].shape[0] != 1:
            raise ValueError("the incremental mode needs batch size 1 (frame t depends on the TSDF after frame t-1)")
        if self.frames > 0:
            if timer is not None:
                timer.start("hint_time")
            if self.fused_hint:
                prepare_mesh_hint_fused(fuser, cur_data, H2, W2)
            else:
                prepare_mesh_hint(fuser, self.renderer, cur_data, H2, W2)
            if timer is not None:
                timer.stop("hint_time")
        else:
            ref = cur_data["cam_T_world_b44"]
            empty_hint(cur_data, torch.zeros(1, 1, H2, W2, device=ref.device, dtype=torch.float32))
            if timer is not None:
                timer.skip("hint_time")
        return cur_data, src_data

    @torch.no_grad()
    def finish_frame(self, cur_data, outputs):
        """Fuse the frame's prediction into the scan's TSDF (nearest upsampling / masking as the drivers do) and count it."""
        depth = _depth_for_fusion(outputs, self.fuse_size, self.mask_pred_depth, per_view_mask=True)
        self.fuser.fuse_frames(depth, cur_data["K_full_depth_b44"], cur_data["cam_T_world_b44"], None)
        i = self.frames
        self.frames += 1
        if self.on_frame is not None:
            self.on_frame(i, cur_data, outputs)

    @torch.no_grad()
    def step(self):
        # (without a lookahead a batch is pulled when its frame starts: a lazy loader is never asked for more than the loop
        #  consumes; with one, the batch after the current frame is pulled first)
        item = self._next if self.lookahead is not None else self._pull()
        if item is None:
            return False
        nxt = self._pull() if self.lookahead is not None else None
        timer, lookahead = self.timer, self.lookahead
        cur_data, src_data = self.begin_frame(item)
        owner = None
        if lookahead is not None and nxt is not None:
            owner = getattr(lookahead, "after_volume_of", None)
            if owner is not None:
                owner.after_volume = (lambda nb=nxt: lookahead(*nb))
            else:
                lookahead(*nxt)
        if timer is not None:
            timer.start("model_time")
        try:
            outputs = self.model_fn(cur_data, src_data)
        finally:
            # the one-shot hook belongs to THIS frame: if model_fn raised or never went through the owner's forward, a
            # stale closure over the next batch must not fire on a later, unrelated forward
            stale = None if owner is None else owner.__dict__.pop("after_volume", None)
        if stale is not None:
            stale()  # (model_fn did not consume it: run the lookahead now, as without an owner)
        if timer is not None:
            timer.stop("model_time")
        self.finish_frame(cur_data, outputs)
        self._next = nxt
        return True


def run_incremental_scan(model_fn, fuser, batches, render_hw, fuse_size=None, fused_hint=True, mask_pred_depth=False,
                         on_frame=None, timer: FrameTimer | None = None, lookahead=None):
    """One scan of the incremental (online) mode; batch size 1 (reference test_incremental.py:25).

    batches: iterable of (cur_data, src_data); cur_data carries K_s0_b44 / invK_s0_b44 / cam_T_world_b44 /
    world_T_cam_b44 / K_full_depth_b44 (+ whatever model_fn reads).  The hint entries (depth_hint_b1hw,
    depth_hint_mask_b1hw, depth_hint_mask_b_b1hw, sampled_weights_b1hw) are written into cur_data here.
    fused_hint: marching-cubes soup -> raster -> one back-project/sample/threshold kernel (4 launches) instead of the
    reference-shaped sequence over a merged mesh.  timer: a FrameTimer that receives hint_time / model_time per frame
    (test_incremental.py:205,256-258,274-288).  lookahead: callable(next_cur_data, next_src_data) run before the model of
    the current frame -- work of frame t+1 that does not depend on frame t's result (``matching_lookahead``: its matching
    features) and can fill the chip beside frame t's latency-bound kernels.  Returns the number of frames fused."""
    scan = IncrementalScan(model_fn, fuser, batches, render_hw, fuse_size=fuse_size, fused_hint=fused_hint,
                           mask_pred_depth=mask_pred_depth, on_frame=on_frame, timer=timer, lookahead=lookahead)
    while scan.step():
        pass
    return scan.frames


def run_incremental_scans(scans, in_flight=3, device=None, max_lead="auto"):
    """Several scans of the incremental mode IN FLIGHT on one GPU (round 6).  Inside a scan frame t needs the TSDF after
    frame t-1, so one scan is a chain of latency-bound kernels that leaves most of the chip idle (0.84 of the time at
    512x384); different scans are independent.  ``scans``: ``IncrementalScan`` objects (each with its own fuser; the model --
    and its feature cache, keyed by scan id -- may be shared).  Scan j runs on lane j % in_flight (a HIP stream), one frame
    per scan per turn, round-robin; a lane is only ever given frames of its own scans, so stream order keeps every scan's
    frames in sequence and no cross-scan ordering is needed.  Results per scan are bit-identical to running it alone
    (same kernels, same order; the conv plan objective is process-wide: pass scans built under the one you want).
    max_lead: back-pressure, as parallel.KeyframePipeline (frames the host may be ahead of the GPU; "auto" = in_flight + 1).
    Returns the list of frames fused per scan."""
    scans = list(scans)
    if not scans:
        return []
    if device is None:
        device = scans[0].fuser.tsdf_fuser_pred.tsdf.device if hasattr(scans[0].fuser, "tsdf_fuser_pred") else torch.device("cuda")
    device = torch.device(device)
    lanes = max(1, min(int(in_flight), len(scans)))
    cuda = device.type == "cuda"
    streams = None
    if cuda and lanes > 1:
        from . import hwqueues

        hwqueues.check(lanes)
        caller = torch.cuda.current_stream(device)
        streams = [torch.cuda.Stream(device) for _ in range(lanes)]
        for st in streams:
            st.wait_stream(caller)
    lead = (lanes + 1) if max_lead == "auto" else int(max_lead or 0)
    pending = []
    active = list(range(len(scans)))
    while active:
        still = []
        for j in active:
            if cuda and lead and len(pending) >= lead:
                pending.pop(0).synchronize()
            if streams is None:
                more = scans[j].step()
            else:
                with torch.cuda.stream(streams[j % lanes]):
                    more = scans[j].step()
            if cuda and lead:
                ev = torch.cuda.Event()
                ev.record(streams[j % lanes] if streams is not None else torch.cuda.current_stream(device))
                pending.append(ev)
            if more:
                still.append(j)
        active = still
    if streams is not None:
        cur = torch.cuda.current_stream(device)
        for st in streams:
            cur.wait_stream(st)
    return [s.frames for s in scans]


def _collate_frames(vals):
    """The batch-1 items of k scans (one turn) -> one item of batch k: tensors are concatenated along their leading extent, a
    per-scan string (``scan_id_string``) becomes the list of k strings the model's feature cache takes, per-element string
    lists are concatenated ("frame_id_string": [id] -> [id_0 .. id_k-1]), dicts and other lists recurse (the source form of
    the ids, K lists of one id, becomes K lists of k; a feature pyramid stays a list of levels); anything else must be the
    same for all scans and is passed through."""
    v0 = vals[0]
    if isinstance(v0, torch.Tensor):
        return v0 if len(vals) == 1 else torch.cat(vals, 0)
    if isinstance(v0, str):
        return list(vals)
    if isinstance(v0, dict):
        return {key: _collate_frames([v[key] for v in vals]) for key in v0}
    if isinstance(v0, (list, tuple)):
        if v0 and all(isinstance(e, str) for e in v0):
            return sum((list(v) for v in vals), [])
        return [_collate_frames([v[j] for v in vals]) for j in range(len(v0))]
    return v0


class IncrementalScanBatch:
    """k scans of the incremental mode advanced in lock step, their current frames evaluated by ONE model call per ``step()``
    (round 6).  Where ``run_incremental_scans`` overlaps the scans' latency-bound kernel chains on HIP streams, this batches
    them: hint preparation per scan, one ``model_fn(cur_data_k, src_data_k)`` on the collated batch of k keyframes (at 512x384
    the conv stack and the volume kernel cost 0.70 ms per frame at batch 4 against 0.87 at batch 1), then every scan fuses
    its own element.  The object has the ``step()`` / ``frames`` / ``fuser`` surface of ``IncrementalScan``, so batches go on
    lanes like scans do: ``run_incremental_scans([IncrementalScanBatch(..), IncrementalScanBatch(..)], in_flight=2)``.
    ``scans``: ``IncrementalScan`` objects without a lookahead (their own ``model_fn`` is not used); ``model_fn`` must accept a
    batch of frames from DIFFERENT scans (``DepthModelCVHint`` does: cameras, hints and feature-cache ids are per element).
    Scans may have different lengths: the batch shrinks as they finish.  Per-scan results equal the scan run alone up to the
    batched-vs-single summation order of the conv kernels (1e-6 relative in depth; the TSDF's fp16 thresholds can turn that
    into isolated differing voxels) -- NOT bit for bit, which is what ``run_incremental_scans`` over plain scans offers."""

    def __init__(self, scans, model_fn):
        self.scans = list(scans)
        if any(s.lookahead is not None for s in self.scans):
            raise ValueError("IncrementalScanBatch takes scans without a lookahead")
        self.model_fn = model_fn
        self._active = list(range(len(self.scans)))

    @property
    def frames(self):
        return sum(s.frames for s in self.scans)

    @property
    def fuser(self):
        return self.scans[0].fuser

    @torch.no_grad()
    def step(self):
        frames, still = [], []
        for j in self._active:
            item = self.scans[j].begin_frame()
            if item is not None:
                frames.append((j, item))
                still.append(j)
        self._active = still
        if not frames:
            return False
        cur = _collate_frames([it[0] for _, it in frames])
        src = _collate_frames([it[1] for _, it in frames])
        outputs = self.model_fn(cur, src)
        for e, (j, (cur_j, _)) in enumerate(frames):
            out_j = {k: (v[e:e + 1] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == len(frames) else v)
                     for k, v in outputs.items()}
            self.scans[j].finish_frame(cur_j, out_j)
        return True


def run_incremental_scans_batched(scans, model_fn, max_lead=2):
    """One ``IncrementalScanBatch`` over all ``scans`` driven to the end on torch's current stream (max_lead: turns the host
    may run ahead of the GPU).  Returns the list of frames fused per scan."""
    scans = list(scans)
    if scans:
        run_incremental_scans([IncrementalScanBatch(scans, model_fn)], in_flight=1, max_lead=max_lead,
                              device=scans[0].fuser.tsdf_fuser_pred.tsdf.device if hasattr(scans[0].fuser, "tsdf_fuser_pred") else "cpu")
    return [s.frames for s in scans]


@torch.no_grad()
def hints_from_mesh(mesh, hint_fuser, renderer, cur_data, render_hw, hint_world_T_world_144=None):
    """Second-pass hint maps of a keyframe batch (test_offline_two_pass.py:311-358, test_revisit.py:194-231): depth
    rendered from the finished hint mesh, weights sampled from the hint TSDF per batch element, NO 0.025 cut (commented
    out in the reference, two_pass :354-356), weights zeroed outside the render mask.  One raster launch + one fused
    back-project / sample / mask launch (dt_hint_from_depth_f32 with the cut disabled) per batch element -- round 2 built
    these maps from a dozen torch ops and a Python loop over elements.

    hint_world_T_world_144: rigid transform taking this scan's world frame to the hint mesh's world frame
    (test_revisit.py:112-114 ``first_scan_T_second_scan``); None when both live in the same frame (two-pass)."""
    import ctypes as C

    H2, W2 = render_hw
    L = _abi.lib()
    tsdf = hint_fuser.tsdf_fuser_pred.tsdf
    dev = tsdf.device
    b = cur_data["cam_T_world_b44"].shape[0]
    pose = cur_data["world_T_cam_b44"].to(device=dev, dtype=torch.float32)
    if hint_world_T_world_144 is not None:
        pose = hint_world_T_world_144.to(device=dev, dtype=torch.float32) @ pose   # first_scan_world_T_cam, :202-205
        cam_T_world = torch.inverse(pose)
    else:
        cam_T_world = cur_data["cam_T_world_b44"].to(device=dev, dtype=torch.float32)
    pose = pose.contiguous()
    K = cur_data["K_s0_b44"].clone()
    K[:, 0] /= W2
    K[:, 1] /= H2
    depth, _ = renderer.render(mesh, cam_T_world.clone(), K)
    invK = cur_data["invK_s0_b44"].to(device=dev, dtype=torch.float32).contiguous()
    hint = torch.empty_like(depth)
    mask_f = torch.empty_like(depth)
    mask_b = torch.empty(b, 1, H2, W2, device=dev, dtype=torch.bool)
    weights = torch.empty_like(depth)
    o = (C.c_float * 3)(*[float(v) for v in tsdf.origin.float().tolist()])
    X, Y, Z = tsdf.tsdf_weights.shape
    stream = _abi.current_stream(dev)
    for j in range(b):  # the reference samples element by element too (:334-341)
        _abi.check(L.dt_hint_from_depth_f32(_abi.ptr(depth[j]), _abi.ptr(tsdf.tsdf_weights), o, float(tsdf.voxel_size), X, Y, Z,
                                            _abi.ptr(invK[j]), _abi.ptr(pose[j]), float("-inf"), H2, W2, _abi.ptr(hint[j]),
                                            _abi.ptr(mask_f[j]), _abi.ptr(mask_b[j]), _abi.ptr(weights[j]),
                                            int(_tsdf_mod.SAMPLE_FP16_MATH), stream),
                   "dt_hint_from_depth_f32")
    cur_data["depth_hint_b1hw"] = hint
    cur_data["depth_hint_mask_b_b1hw"] = mask_b
    cur_data["depth_hint_mask_b1hw"] = mask_f
    cur_data["sampled_weights_b1hw"] = weights
    return depth


def two_pass_fns(model_fn, load_batch, render_hw, fuse_size=None, mask_pred_depth=False, on_frame=None,
                 load_first_pass_batch=None, hint_world_T_world_144=None, timer: FrameTimer | None = None):
    """(first_pass_fn, between_passes, second_pass_fn) for ``parallel.run_two_pass`` (``in_flight=4`` there runs the keyframe
    batches of both passes through ``parallel.KeyframePipeline``; with ``model.enable_launch_programs()`` the outputs of
    ``model_fn`` are lane-static buffers, which the step functions below consume before they return).

    ``load_batch(i) -> (cur_data, src_data)`` fetches keyframe batch i onto this rank's GPU (only called for the
    rank's own batches).  Second-pass hints: ``hints_from_mesh``.  ``load_first_pass_batch`` / ``hint_world_T_world_144``
    serve ``revisit_fns`` (first pass over another scan; hints through a rigid transform).  timer: second-pass
    hint_time / model_time per frame (test_offline_two_pass.py:313,360-362,371-377)."""
    H2, W2 = render_hw
    load_first = load_first_pass_batch if load_first_pass_batch is not None else load_batch

    empty = {}  # (b, device) -> the pass's empty hint maps: constants the model only reads, built once, not per batch

    @torch.no_grad()
    def first(i):
        cur_data, src_data = load_first(i)
        b = cur_data["cam_T_world_b44"].shape[0]
        dev = cur_data["cam_T_world_b44"].device
        maps = empty.get((b, dev))
        if maps is None:
            maps = {}
            empty_hint(maps, torch.zeros(b, 1, H2, W2, device=dev, dtype=torch.float32))
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()  # (read from every lane of a KeyframePipeline afterwards)
            empty[(b, dev)] = maps
        cur_data.update(maps)
        out = model_fn(cur_data, src_data)
        return _depth_for_fusion(out, fuse_size, mask_pred_depth, per_view_mask=False), cur_data["K_full_depth_b44"], \
            cur_data["cam_T_world_b44"]

    def between(hint_fuser):
        mesh, _, _ = hint_fuser.get_mesh_pytorch3d(scale_to_world=True)  # replicas are identical: local, no collective
        return dict(mesh=mesh, hint_fuser=hint_fuser, renderer=MeshDepthRenderer(H2, W2))

    @torch.no_grad()
    def second(i, state):
        cur_data, src_data = load_batch(i)
        b = cur_data["cam_T_world_b44"].shape[0]
        if timer is not None:
            timer.start("hint_time")
        hints_from_mesh(state["mesh"], state["hint_fuser"], state["renderer"], cur_data, (H2, W2), hint_world_T_world_144)
        if timer is not None:
            timer.stop("hint_time", b)
            timer.start("model_time")
        out = model_fn(cur_data, src_data)
        if timer is not None:
            timer.stop("model_time", b)
        if on_frame is not None:
            on_frame(i, cur_data, out)
        return _depth_for_fusion(out, fuse_size, mask_pred_depth, per_view_mask=False), cur_data["K_full_depth_b44"], \
            cur_data["cam_T_world_b44"]

    return first, between, second


def revisit_fns(model_fn, load_first_scan_batch, load_batch, first_scan_T_second_scan_144, render_hw, fuse_size=None,
                mask_pred_depth=False, on_frame=None, timer: FrameTimer | None = None):
    """The revisit flow of reference test_revisit.py:104-260 as the step functions of ``parallel.run_two_pass``
    (``num_first_batches=`` gives the first scan's batch count):

      first(i)          one keyframe batch of the PREVIOUS scan with empty hints (compute_hint_mesh, :122-156 -> the
                        hint TSDF at 0.04 m / 3 m that ``run_two_pass`` fuses, like the two-pass first pass)
      between(fuser)    marching cubes of that TSDF, once
      second(i, state)  one keyframe batch of the NEW scan: hints rendered from the previous scan's mesh with the camera
                        ``inverse(first_scan_T_second_scan @ world_T_cam)`` and weights sampled at
                        ``first_scan_T_second_scan @ world_T_cam @ cam_points`` (:194-231), then the model; the returned
                        depth / K / cam_T_world fuse into the NEW scan's volume in its own world frame (:262-316).

    ``first_scan_T_second_scan_144``: [1,4,4], the inverse of the rescan transform the 3RScan metadata lists (:112-114)."""
    return two_pass_fns(model_fn, load_batch, render_hw, fuse_size=fuse_size, mask_pred_depth=mask_pred_depth,
                        on_frame=on_frame, load_first_pass_batch=load_first_scan_batch,
                        hint_world_T_world_144=first_scan_T_second_scan_144, timer=timer)
