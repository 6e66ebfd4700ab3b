"""Voxel-block (sparse) fuser with the reference's ``CustomOpen3dFuser`` interface
(reference tools/fusers_helper.py:263-511), on the kernels of csrc/sparse_tsdf.hip.

The reference class wraps Open3D's ``VoxelBlockGrid`` (open3d==0.18.0, third party; not installed here -- its published
block-activation / meshing behaviour is restated, PARITY UNPINNED) and does the TSDF update itself with Open3D tensor ops
(:369-441), which IS restated exactly (fp32).  Layout: a dense block directory + an append-only pool of 16^3 fp32 tiles
(see the kernel file for why a directory instead of a hash map on a 288 GB part).

Not built: colour fusion (``fuse_color`` is accepted and ignored with a warning -- the drivers only use it for
visualisation), trimesh / Open3D mesh objects (``get_mesh`` returns a vertices/faces holder).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _abi
from ..utils.pytorch3d_extras import merge_by_edge_id
from .fusers_helper import DepthFuser, SimpleMesh
from .tsdf import Meshes


class SparseVoxelBlockGrid:
    """The seven device buffers of a voxel-block grid (include/doubletake_hip.h, "voxel-block (sparse) fp32 TSDF")."""

    BLOCK = 16

    def __init__(self, voxel_size, device, capacity=32768, half_extent_m=20.48):
        self.voxel_size = float(voxel_size)
        self.device = device
        block_m = self.voxel_size * self.BLOCK
        nb = int(np.ceil(half_extent_m / block_m)) * 2
        self.nb = max(2, min(nb, 1024))
        self.capacity = int(capacity)
        n = self.nb ** 3
        self.dir = torch.full((n,), -1, dtype=torch.int32, device=device)
        self.touch = torch.zeros(n, dtype=torch.uint8, device=device)
        self.keys = torch.zeros(self.capacity * 3, dtype=torch.int32, device=device)
        self.tsdf = torch.zeros(self.capacity * self.BLOCK ** 3, dtype=torch.float32, device=device)
        self.weight = torch.zeros_like(self.tsdf)
        self.count = torch.zeros(2, dtype=torch.int32, device=device)

    def args(self):
        return (_abi.ptr(self.dir), _abi.ptr(self.touch), self.nb, self.voxel_size, _abi.ptr(self.keys), _abi.ptr(self.tsdf),
                _abi.ptr(self.weight), _abi.ptr(self.count), self.capacity)

    def num_blocks(self, check=True):
        """Allocated blocks (one host read).  Raises if any block could not be placed."""
        used, lost = (int(v) for v in self.count.tolist())
        if check and lost:
            raise _abi.DoubletakeHipError(
                f"sparse TSDF: {lost} block activations did not fit (capacity {self.capacity} blocks, directory +-"
                f"{self.nb // 2 * self.BLOCK * self.voxel_size:.1f} m); construct the fuser with a larger capacity / extent")
        return used

    def grow(self, new_capacity):
        """Enlarge the pool (contents kept).  288 GB of HBM: doubling is cheap."""
        extra = int(new_capacity) - self.capacity
        if extra <= 0:
            return
        v = self.BLOCK ** 3
        self.keys = torch.cat([self.keys, torch.zeros(extra * 3, dtype=torch.int32, device=self.device)])
        self.tsdf = torch.cat([self.tsdf, torch.zeros(extra * v, dtype=torch.float32, device=self.device)])
        self.weight = torch.cat([self.weight, torch.zeros(extra * v, dtype=torch.float32, device=self.device)])
        self.capacity = int(new_capacity)

    def block_keys(self):
        n = self.num_blocks()
        return self.keys[: n * 3].view(n, 3)


class CustomOpen3dFuser(DepthFuser):
    """Reference tools/fusers_helper.py:263-511.  Constructor keywords as the reference's (gt_path is accepted and
    unused there too: a voxel-block grid needs no bounds)."""

    CHECK_EVERY = 16  # frames between read-backs of the allocation counter

    def __init__(self, extended_neg_truncation=False, gt_path="", fusion_resolution=0.04, max_fusion_depth=3.0,
                 fuse_color=False, capacity=65536, half_extent_m=20.48, device=None):
        super().__init__(gt_path, fusion_resolution, max_fusion_depth, fuse_color)
        if not torch.cuda.is_available():
            raise _abi.DoubletakeHipError("the sparse fuser needs a ROCm GPU (no CPU fallback)")
        if fuse_color:
            print("WARNING: colour fusion is not built in doubletake_amd's voxel-block fuser; fusing geometry only.")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.sdf_trunc = 3 * self.fusion_resolution
        self.weight_threshold = 0.00000001
        self.extended_neg_truncation = extended_neg_truncation
        self.volume = SparseVoxelBlockGrid(self.fusion_resolution, self.device, capacity, half_extent_m)
        self.frames_fused = 0

    @torch.no_grad()
    def fuse_frames(self, depths_b1hw, K_b44, cam_T_world_b44, color_b3hw=None, hint_b1hw=None):
        L = _abi.lib()
        g = self.volume
        dev = g.device
        depth = depths_b1hw.to(device=dev, dtype=torch.float32).contiguous()
        # cameras stay on the device (dt_sparse_integrate_frames_f32 reads them there): no .cpu() -- i.e. no device
        # synchronisation -- per call, one entry-point call per CHECK_EVERY-aligned run of frames
        K = K_b44.detach().to(device=dev, dtype=torch.float32).contiguous()
        T = cam_T_world_b44.detach().to(device=dev, dtype=torch.float32).contiguous()
        n, (h, w) = depth.shape[0], depth.shape[-2:]
        stream = _abi.current_stream(dev)
        i = 0
        while i < n:
            # the pool is sized generously (288 GB of HBM); the allocation counter is read back every CHECK_EVERY frames
            # (and at every mesh extraction): the pool doubles once it is half full, and a block that did not fit is never
            # silent -- num_blocks() raises
            if self.frames_fused % self.CHECK_EVERY == self.CHECK_EVERY - 1:
                used = g.num_blocks()
                if used * 2 > g.capacity:
                    g.grow(2 * g.capacity)
            run = min(n - i, self.CHECK_EVERY - (self.frames_fused + 1) % self.CHECK_EVERY) if self.CHECK_EVERY > 1 else 1
            run = max(1, run)
            _abi.check(L.dt_sparse_integrate_frames_f32(*g.args(), _abi.ptr(depth[i:i + run]), run, h, w, _abi.ptr(K[i:i + run]),
                                                        _abi.ptr(T[i:i + run]), float(self.max_fusion_depth), 3.0,
                                                        int(bool(self.extended_neg_truncation)), stream),
                       "dt_sparse_integrate_frames_f32")
            self.frames_fused += run
            i += run

    @torch.no_grad()
    def _extract(self, weight_threshold=None, trim_tsdf_using_confience=False):
        """Triangle soup of the zero level set: (verts [V,3] world, vertex weights [V], faces [V/3,3], edge ids [V])."""
        L = _abi.lib()
        g = self.volume
        dev = g.device
        n = g.num_blocks()
        wthr = self.weight_threshold if weight_threshold is None else weight_threshold
        if trim_tsdf_using_confience:  # reference :459-460 zeroes the sdf of low-confidence voxels before meshing
            g.tsdf[g.weight < 0.02] = 0
        empty = (torch.zeros(0, 3, device=dev), torch.zeros(0, device=dev), torch.zeros(0, 3, dtype=torch.int64, device=dev),
                 torch.zeros(0, dtype=torch.int64, device=dev))
        if n == 0:
            return empty
        stream = _abi.current_stream(dev)
        offsets = torch.empty(n, dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        _abi.check(L.dt_sparse_mc_count(*g.args(), n, 0.0, float(wthr), _abi.ptr(offsets), _abi.ptr(total), stream),
                   "dt_sparse_mc_count")
        nverts = int(total.item())
        if nverts < 0:
            raise _abi.DoubletakeHipError("sparse marching cubes: vertex count overflows int32")
        if nverts == 0:
            return empty
        verts = torch.empty(nverts, 3, dtype=torch.float32, device=dev)
        vw = torch.empty(nverts, dtype=torch.float32, device=dev)
        faces = torch.empty(nverts // 3, 3, dtype=torch.int64, device=dev)
        ids = torch.empty(nverts, dtype=torch.int64, device=dev)
        _abi.check(L.dt_sparse_mc_generate(*g.args(), n, 0.0, float(wthr), _abi.ptr(offsets), _abi.ptr(verts), _abi.ptr(vw),
                                           _abi.ptr(faces), _abi.ptr(ids), nverts, stream), "dt_sparse_mc_generate")
        return verts, vw, faces, ids

    def get_mesh_pytorch3d(self, scale_to_world=True):
        """Reference :483-505 -> (mesh, verts, faces); the mesh's ``textures`` holds the per-vertex fused weight
        (the reference stores the confidence in the red channel of the vertex colours, :464-470)."""
        verts, vw, faces, ids = self._extract()
        if verts.shape[0] == 0:
            verts = torch.zeros(1, 3, device=self.volume.device)
            faces = torch.zeros(1, 3, dtype=torch.int64, device=self.volume.device)
            return Meshes(verts=[verts], faces=[faces], textures=torch.zeros(1, device=verts.device)), verts, faces
        packed = torch.cat([verts, vw[:, None]], 1)
        merged, faces = merge_by_edge_id(packed, faces, ids)
        verts = merged[:, :3].contiguous()
        if not scale_to_world:
            verts = verts / self.volume.voxel_size
        return Meshes(verts=[verts], faces=[faces], textures=merged[:, 3].contiguous()), verts, faces

    def get_mesh(self, export_single_mesh=None, convert_to_trimesh=False, get_confidence=False, trim_tsdf_using_confience=False):
        verts, vw, faces, ids = self._extract(trim_tsdf_using_confience=trim_tsdf_using_confience)
        if verts.shape[0] == 0:
            return SimpleMesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64))
        merged, faces = merge_by_edge_id(verts, faces, ids)
        return SimpleMesh(merged.cpu().numpy(), faces.cpu().numpy())

    def export_mesh(self, path, use_marching_cubes_mask=None, trim_tsdf_using_confience=False):
        from ..utils.formats import write_ply

        m = self.get_mesh(trim_tsdf_using_confience=trim_tsdf_using_confience)
        write_ply(path, m.vertices, m.faces, comment="doubletake_amd: voxel-block marching cubes (CustomOpen3dFuser)")

    def save_tsdf(self, path):
        """Reference :211-212 is a no-op for this fuser; here the block pool is saved (keys + tiles) for debugging."""
        n = self.volume.num_blocks()
        v = SparseVoxelBlockGrid.BLOCK
        np.savez_compressed(path, keys=self.volume.block_keys().cpu().numpy(), voxel_size=self.volume.voxel_size,
                            tsdf=self.volume.tsdf[: n * v ** 3].view(n, v, v, v).cpu().numpy(),
                            weight=self.volume.weight[: n * v ** 3].view(n, v, v, v).cpu().numpy())

    @torch.no_grad()
    def sample_tsdf(self, world_points_N3, what_to_sample="tsdf", sampling_method="bilinear"):
        """Trilinear sample of the sparse field (extension: the reference reads the confidence of a hint pixel from the
        rendered vertex colours instead, test_incremental.py:238-241)."""
        if what_to_sample not in ("tsdf", "weights"):
            raise ValueError(what_to_sample)
        L = _abi.lib()
        g = self.volume
        pts = world_points_N3.to(device=g.device, dtype=torch.float32).contiguous()
        out = torch.empty(pts.shape[0], dtype=torch.float32, device=g.device)
        _abi.check(L.dt_sparse_sample_f32(*g.args(), _abi.ptr(pts), _abi.ptr(out), pts.shape[0], 1 if what_to_sample == "weights" else 0,
                                          _abi.current_stream(g.device)), "dt_sparse_sample_f32")
        return out
