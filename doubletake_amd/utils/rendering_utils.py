"""Hint-mesh depth renderer with the reference's class interface
(reference utils/rendering_utils.py:9-53, PyTorch3DMeshDepthRenderer) on the HIP rasteriser of
csrc/raster.hip, plus the hint-preparation block of the incremental driver
(reference test_incremental.py:187-269) as one function.
"""
from __future__ import annotations

import torch

from .. import _abi
from ..tools import tsdf as _tsdf_mod


class MeshDepthRenderer:
    """render(mesh, cam_T_world_b44, K_b44) -> (depth_b1hw, None); K is *normalised* (row 0 / width,
    row 1 / height) exactly as the reference passes it.  `mesh` is anything with verts_list()/faces_list()
    (doubletake_amd.tools.tsdf.Meshes, or a PyTorch3D Meshes)."""

    def __init__(self, height=192, width=256) -> None:
        self.height = height
        self.width = width

    @torch.no_grad()
    def render(self, mesh, cam_T_world_b44, K_b44, render_color=False):
        if render_color:
            raise NotImplementedError("colour rendering is only used with the open3d fuser (out of scope)")
        L = _abi.lib()
        verts = mesh.verts_list()[0]
        faces = mesh.faces_list()[0]
        dev = cam_T_world_b44.device
        if not dev.type == "cuda":
            raise _abi.DoubletakeHipError("the mesh renderer only runs on a ROCm GPU (no CPU fallback)")
        verts = verts.to(device=dev, dtype=torch.float32).contiguous()
        faces = faces.to(device=dev, dtype=torch.int64).contiguous()
        nf = int(faces.shape[0]) if faces.dim() == 2 and verts.shape[0] >= 3 else 0
        b = cam_T_world_b44.shape[0]
        K = K_b44.to(dev).float().clone()
        K[:, 0] *= self.width
        K[:, 1] *= self.height
        T = cam_T_world_b44.to(dev).float().contiguous()
        K = K.contiguous()
        out = torch.empty(b, 1, self.height, self.width, device=dev, dtype=torch.float32)
        ws = torch.empty(self.height * self.width, device=dev, dtype=torch.int32)
        stream = _abi.current_stream(dev)
        for i in range(b):
            _abi.check(L.dt_raster_depth_f32(_abi.ptr(verts), _abi.ptr(faces), nf, _abi.ptr(T[i]), _abi.ptr(K[i]), self.height,
                                             self.width, _abi.ptr(ws), _abi.ptr(out[i, 0]), stream), "dt_raster_depth_f32")
        return out, None


PyTorch3DMeshDepthRenderer = MeshDepthRenderer  # reference name


@torch.no_grad()
def prepare_mesh_hint(fuser, mesh_renderer, cur_data, render_height, render_width, weight_threshold=0.025):
    """test_incremental.py:187-258: mesh -> rendered depth -> sampled TSDF weights -> hint dict entries.
    Writes depth_hint_b1hw / depth_hint_mask_b1hw / depth_hint_mask_b_b1hw / sampled_weights_b1hw into
    cur_data (batch size 1, as the incremental driver requires) and returns the rendered depth."""
    mesh, _, _ = fuser.get_mesh_pytorch3d(scale_to_world=True)
    K = cur_data["K_s0_b44"].clone()
    K[:, 0] /= render_width
    K[:, 1] /= render_height
    depth, _ = mesh_renderer.render(mesh, cur_data["cam_T_world_b44"].clone(), K)
    hint = depth.clone()
    hint[hint == -1] = float("nan")
    # backproject (pixel centres at +0.5, utils/geometry_utils.py:34-39) and move to world
    dev = depth.device
    ys, xs = torch.meshgrid(torch.arange(render_height, device=dev), torch.arange(render_width, device=dev), indexing="ij")
    pix = torch.stack([xs.flatten() + 0.5, ys.flatten() + 0.5, torch.ones(render_height * render_width, device=dev)], 0)
    cam = (cur_data["invK_s0_b44"][0, :3, :3].float() @ pix) * depth.reshape(1, -1)
    cam4 = torch.cat([cam, torch.ones_like(cam[:1])], 0)
    world = (cur_data["world_T_cam_b44"][0].float() @ cam4)[:3].t().contiguous()
    weights = fuser.sample_tsdf(world, what_to_sample="weights").view(1, 1, render_height, render_width)
    hint[weights < weight_threshold] = float("nan")
    mask_b = ~torch.isnan(hint)
    weights = weights.clone()
    weights[~mask_b] = 0.0
    cur_data["depth_hint_b1hw"] = hint
    cur_data["depth_hint_mask_b_b1hw"] = mask_b
    cur_data["depth_hint_mask_b1hw"] = mask_b.float()
    cur_data["sampled_weights_b1hw"] = weights
    return depth


@torch.no_grad()
def prepare_mesh_hint_fused(fuser, cur_data, render_height, render_width, weight_threshold=0.025, via_soup=False,
                            fp16_math=None):
    """Same outputs as prepare_mesh_hint without an intermediate mesh object and without a host read: marching cubes
    fused with the depth render (dt_mc_raster_depth_f32: z-buffer init, one kernel that compacts each workgroup's
    triangles in LDS and rasterises them, resolve) -> back-projection + weight sampling + threshold.  The rendered image
    can differ from prepare_mesh_hint's in isolated pixels (vertices are not merged by edge id, so a shared vertex may
    come from either neighbouring cell's interpolation: <= 1 ulp apart).
    via_soup=True: the round-2 sequence (count -> scan -> host read of the vertex count -> generate -> raster), kept for
    the bit-equality test of the two."""
    import ctypes as C

    L = _abi.lib()
    tsdf = fuser.tsdf_fuser_pred.tsdf
    dev = tsdf.device
    stream = _abi.current_stream(dev)
    h, w = render_height, render_width
    o = (C.c_float * 3)(*[float(v) for v in tsdf.origin.float().tolist()])
    T = cur_data["cam_T_world_b44"][0].to(device=dev, dtype=torch.float32).contiguous()
    K = cur_data["K_s0_b44"][0].to(device=dev, dtype=torch.float32).contiguous()
    depth = torch.empty(1, 1, h, w, device=dev, dtype=torch.float32)
    ws = torch.empty(h * w, device=dev, dtype=torch.int32)
    if via_soup:
        from .pytorch3d_extras import marching_cubes_raw

        verts, _, _ = marching_cubes_raw(tsdf.tsdf_values, tsdf.voxel_bitmap, 0.0)
        nf = int(verts.shape[0]) // 3
        _abi.check(L.dt_raster_soup_depth_f32(_abi.ptr(verts), nf, o, float(tsdf.voxel_size), _abi.ptr(T), _abi.ptr(K), h, w,
                                              _abi.ptr(ws), _abi.ptr(depth), stream), "dt_raster_soup_depth_f32")
    else:
        vol = tsdf.tsdf_values
        if vol.dtype != torch.float16 or not vol.is_contiguous():
            vol = vol.half().contiguous()
        X, Y, Z = vol.shape
        _abi.check(L.dt_mc_raster_depth_f32(_abi.ptr(vol), _abi.ptr(tsdf.voxel_bitmap), X, Y, Z, 0.0, None, None, o,
                                            float(tsdf.voxel_size), _abi.ptr(T), _abi.ptr(K), h, w, _abi.ptr(ws),
                                            _abi.ptr(depth), stream), "dt_mc_raster_depth_f32")
    invK = cur_data["invK_s0_b44"][0].to(device=dev, dtype=torch.float32).contiguous()
    pose = cur_data["world_T_cam_b44"][0].to(device=dev, dtype=torch.float32).contiguous()
    hint = torch.empty_like(depth)
    mask_f = torch.empty_like(depth)
    mask_b = torch.empty(1, 1, h, w, device=dev, dtype=torch.bool)
    weights = torch.empty_like(depth)
    X, Y, Z = tsdf.tsdf_weights.shape
    _abi.check(L.dt_hint_from_depth_f32(_abi.ptr(depth), _abi.ptr(tsdf.tsdf_weights), o, float(tsdf.voxel_size), X, Y, Z,
                                        _abi.ptr(invK), _abi.ptr(pose), float(weight_threshold), h, w, _abi.ptr(hint),
                                        _abi.ptr(mask_f), _abi.ptr(mask_b), _abi.ptr(weights),
                                        int(_tsdf_mod.SAMPLE_FP16_MATH if fp16_math is None else fp16_math), stream),
               "dt_hint_from_depth_f32")
    cur_data["depth_hint_b1hw"] = hint
    cur_data["depth_hint_mask_b_b1hw"] = mask_b
    cur_data["depth_hint_mask_b1hw"] = mask_f
    cur_data["sampled_weights_b1hw"] = weights
    return depth


def empty_hint(cur_data, like_b1hw):
    """test_incremental.py:260-269: all-NaN hint, zero mask and weights (first frame / pass 1)."""
    cur_data["depth_hint_b1hw"] = torch.full_like(like_b1hw, float("nan"))
    cur_data["depth_hint_mask_b1hw"] = torch.zeros_like(like_b1hw)
    cur_data["depth_hint_mask_b_b1hw"] = cur_data["depth_hint_mask_b1hw"].bool()
    cur_data["sampled_weights_b1hw"] = torch.zeros_like(like_b1hw)
