"""Per-scan evaluation loops around the hot path, with the data contracts of the reference drivers.

``run_incremental_scan``  -- reference test_incremental.py:172-372: frame t's hint is rendered from the TSDF fused
                            from frames < t (marching cubes -> depth render -> weight sampling), then model, then fuse.
``two_pass_fns``          -- reference test_offline_two_pass.py:26-131 (first pass, empty hints, hint TSDF at
                            0.04 m / 3 m) and :292-500 (second pass, hints from the finished first-pass mesh) as the
                            two step functions ``parallel.run_two_pass`` shards over GPUs.

The dataset / dataloader, metric averaging, visualisation and file output of those scripts are out of scope
(SURVEY.md section 2); a batch here is the pair of dicts ``(cur_data, src_data)`` the reference dataloaders yield,
already on the GPU.  ``model_fn(cur_data, src_data) -> outputs`` is ``lambda c, s: model("test", c, s,
return_mask=True)`` for a DepthModelCVHint with an image encoder attached, or any callable with that contract.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .utils.rendering_utils import MeshDepthRenderer, empty_hint, prepare_mesh_hint, prepare_mesh_hint_fused


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
def run_incremental_scan(model_fn, fuser, batches, render_hw, fuse_size=None, fused_hint=True, mask_pred_depth=False,
                         on_frame=None):
    """One scan of the incremental (online) mode; batch size 1 (reference test_incremental.py:25).

    batches: iterable of (cur_data, src_data); cur_data carries K_s0_b44 / invK_s0_b44 / cam_T_world_b44 /
    world_T_cam_b44 / K_full_depth_b44 (+ whatever model_fn reads).  The hint entries (depth_hint_b1hw,
    depth_hint_mask_b1hw, depth_hint_mask_b_b1hw, sampled_weights_b1hw) are written into cur_data here.
    fused_hint: marching-cubes soup -> raster -> one back-project/sample/threshold kernel (4 launches) instead of the
    reference-shaped sequence over a merged mesh.  Returns the number of frames fused."""
    H2, W2 = render_hw
    renderer = None if fused_hint else MeshDepthRenderer(H2, W2)
    n = 0
    for i, (cur_data, src_data) in enumerate(batches):
        if cur_data["cam_T_world_b44"].shape[0] != 1:
            raise ValueError("the incremental mode needs batch size 1 (frame t depends on the TSDF after frame t-1)")
        if i > 0:
            if fused_hint:
                prepare_mesh_hint_fused(fuser, cur_data, H2, W2)
            else:
                prepare_mesh_hint(fuser, renderer, cur_data, H2, W2)
        else:
            ref = cur_data["cam_T_world_b44"]
            empty_hint(cur_data, torch.zeros(1, 1, H2, W2, device=ref.device, dtype=torch.float32))
        outputs = model_fn(cur_data, src_data)
        depth = _depth_for_fusion(outputs, fuse_size, mask_pred_depth, per_view_mask=True)
        fuser.fuse_frames(depth, cur_data["K_full_depth_b44"], cur_data["cam_T_world_b44"], None)
        n += 1
        if on_frame is not None:
            on_frame(i, cur_data, outputs)
    return n


def two_pass_fns(model_fn, load_batch, render_hw, fuse_size=None, mask_pred_depth=False, on_frame=None):
    """(first_pass_fn, between_passes, second_pass_fn) for ``parallel.run_two_pass``.

    ``load_batch(i) -> (cur_data, src_data)`` fetches keyframe batch i onto this rank's GPU (only called for the
    rank's own batches).  Second-pass hints follow test_offline_two_pass.py:311-358: depth rendered from the
    first-pass mesh, weights sampled from the first-pass TSDF per batch element, NO 0.025 cut (commented out in the
    reference, :354-356), weights zeroed outside the render mask."""
    H2, W2 = render_hw

    @torch.no_grad()
    def first(i):
        cur_data, src_data = load_batch(i)
        b = cur_data["cam_T_world_b44"].shape[0]
        empty_hint(cur_data, torch.zeros(b, 1, H2, W2, device=cur_data["cam_T_world_b44"].device, dtype=torch.float32))
        out = model_fn(cur_data, src_data)
        return _depth_for_fusion(out, fuse_size, mask_pred_depth, per_view_mask=False), cur_data["K_full_depth_b44"], \
            cur_data["cam_T_world_b44"]

    def between(hint_fuser):
        mesh, _, _ = hint_fuser.get_mesh_pytorch3d(scale_to_world=True)  # replicas are identical: local, no collective
        return dict(mesh=mesh, hint_fuser=hint_fuser, renderer=MeshDepthRenderer(H2, W2))

    @torch.no_grad()
    def second(i, state):
        cur_data, src_data = load_batch(i)
        dev = cur_data["cam_T_world_b44"].device
        b = cur_data["cam_T_world_b44"].shape[0]
        K = cur_data["K_s0_b44"].clone()
        K[:, 0] /= W2
        K[:, 1] /= H2
        depth, _ = state["renderer"].render(state["mesh"], cur_data["cam_T_world_b44"].clone(), K)
        hint = depth.clone()
        hint[hint == -1] = float("nan")
        mask_b = ~torch.isnan(hint)
        ys, xs = torch.meshgrid(torch.arange(H2, device=dev), torch.arange(W2, device=dev), indexing="ij")
        pix = torch.stack([xs.flatten() + 0.5, ys.flatten() + 0.5, torch.ones(H2 * W2, device=dev)], 0)
        weights = []
        for j in range(b):  # :334-341 samples element by element
            cam = (cur_data["invK_s0_b44"][j, :3, :3].float() @ pix) * depth[j].reshape(1, -1)
            world = (cur_data["world_T_cam_b44"][j].float() @ torch.cat([cam, torch.ones_like(cam[:1])], 0))[:3].t().contiguous()
            weights.append(state["hint_fuser"].sample_tsdf(world, what_to_sample="weights"))
        weights = torch.stack(weights, 0).view(b, 1, H2, W2).clone()
        weights[~mask_b] = 0.0
        cur_data["depth_hint_b1hw"] = hint
        cur_data["depth_hint_mask_b_b1hw"] = mask_b
        cur_data["depth_hint_mask_b1hw"] = mask_b.float()
        cur_data["sampled_weights_b1hw"] = weights
        out = model_fn(cur_data, src_data)
        if on_frame is not None:
            on_frame(i, cur_data, out)
        return _depth_for_fusion(out, fuse_size, mask_pred_depth, per_view_mask=False), cur_data["K_full_depth_b44"], \
            cur_data["cam_T_world_b44"]

    return first, between, second
