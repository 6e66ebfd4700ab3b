"""ORACLE (test infrastructure, NOT product code) -- numpy fp32 restatement of the reference's voxel-block fuser
(CustomOpen3dFuser, tools/fusers_helper.py:263-511).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED for the parts that live in Open3D (open3d==0.18.0, environment.yml:37; not installed, absent from
/root/reference): block activation = VoxelBlockGrid.compute_unique_block_coordinates (DepthTouch: stride-4 pixel lattice,
4 samples from max(d - trunc, 0) to min(d + trunc, depth_max) along the ray through pixel index (x, y), block =
floor(point / (16 * voxel_size))) and voxel_coordinates (voxel corner = (key * 16 + local) * voxel_size) are restated
from the published algorithm.  The TSDF update itself (update_tsdf_for_voxels, :369-441) is the reference's own tensor
code and is restated line by line.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
RES = 16


class SparseVolume:
    def __init__(self, voxel_size):
        self.voxel_size = F32(voxel_size)
        self.blocks = {}  # (bx,by,bz) -> [tsdf 16^3, weight 16^3]

    def activate(self, keys):
        for k in keys:
            if k not in self.blocks:
                self.blocks[k] = [np.zeros((RES,) * 3, F32), np.zeros((RES,) * 3, F32)]


def touched_blocks(depth_hw, K44, cam_T_world44, voxel_size, depth_max, trunc):
    """compute_unique_block_coordinates(depth, K, extrinsic, depth_scale=1, depth_max, trunc_voxel_multiplier) -> set."""
    K = np.asarray(K44, F32)
    T = np.asarray(cam_T_world44, F32)
    Rinv = T[:3, :3].T.copy()
    c = -(Rinv @ T[:3, 3])
    H, W = depth_hw.shape
    ys, xs = np.meshgrid(np.arange(0, (H // 4) * 4, 4), np.arange(0, (W // 4) * 4, 4), indexing="ij")
    d = depth_hw[ys, xs].astype(F32)
    ok = (d > 0) & (d < F32(depth_max))
    xc = ((xs.astype(F32) - K[0, 2]) / K[0, 0]).astype(F32)
    yc = ((ys.astype(F32) - K[1, 2]) / K[1, 1]).astype(F32)
    dirs = np.stack([Rinv[i, 0] * xc + Rinv[i, 1] * yc + Rinv[i, 2] for i in range(3)], -1).astype(F32)
    t_min = np.maximum(d - F32(trunc), F32(0))
    t_max = np.minimum(d + F32(trunc), F32(depth_max))
    step = ((t_max - t_min) / F32(3)).astype(F32)
    inv_bs = F32(1.0) / (F32(voxel_size) * F32(RES))
    keys = set()
    t = t_min.copy()
    for s in range(4):
        p = (c[None, None, :].astype(F32) + t[..., None] * dirs).astype(F32)
        b = np.floor(p * inv_bs).astype(np.int64)
        for k in b[ok]:
            keys.add((int(k[0]), int(k[1]), int(k[2])))
        t = (t + step).astype(F32)
    return keys


def integrate(vol: SparseVolume, depth_hw, K44, cam_T_world44, max_depth, extended_neg_truncation=False):
    """fuse_frames for one frame (:326-366) + update_tsdf_for_voxels (:369-441) over ALL allocated blocks."""
    vs = vol.voxel_size
    trunc = F32(3) * vs
    vol.activate(touched_blocks(depth_hw, K44, cam_T_world44, vs, max_depth, trunc))
    K = np.asarray(K44, F32)[:3, :3]
    T = np.asarray(cam_T_world44, F32)
    R, tt = T[:3, :3], T[:3, 3]
    H, W = depth_hw.shape
    depth = depth_hw.astype(F32)
    l = np.arange(RES)
    lx, ly, lz = np.meshgrid(l, l, l, indexing="ij")
    min_sdf = -trunc * F32(1.5) if extended_neg_truncation else -trunc
    for (bx, by, bz), (ts, ws) in vol.blocks.items():
        wx = ((bx * RES + lx).astype(F32) * vs).astype(F32)
        wy = ((by * RES + ly).astype(F32) * vs).astype(F32)
        wz = ((bz * RES + lz).astype(F32) * vs).astype(F32)
        cam = [(R[i, 0] * wx + R[i, 1] * wy + R[i, 2] * wz + tt[i]).astype(F32) for i in range(3)]
        pu = (K[0, 0] * cam[0] + K[0, 1] * cam[1] + K[0, 2] * cam[2]).astype(F32)
        pv = (K[1, 0] * cam[0] + K[1, 1] * cam[1] + K[1, 2] * cam[2]).astype(F32)
        pz = (K[2, 0] * cam[0] + K[2, 1] * cam[1] + K[2, 2] * cam[2]).astype(F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            fx = np.sign(pu / pz) * np.floor(np.abs(pu / pz) + F32(0.5))  # round half away from zero
            fy = np.sign(pv / pz) * np.floor(np.abs(pv / pz) + F32(0.5))
        m = (pz > 0) & (fx >= 0) & (fy >= 0) & (fx < W) & (fy < H)
        xi = np.where(m, fx, 0).astype(np.int64)
        yi = np.where(m, fy, 0).astype(np.int64)
        d = depth[yi, xi]
        sdf = (d - pz).astype(F32)
        inl = m & (d > 0) & (d < F32(max_depth)) & (sdf >= min_sdf)
        sdf = (np.minimum(sdf, trunc) / trunc).astype(F32)
        conf = np.clip(F32(1.0) - (d - F32(0.5)) / (F32(max_depth) - F32(0.5)), F32(0.25), F32(1.0)).astype(F32)
        conf = (conf * conf).astype(F32)
        wn = (conf * F32(2.5) / F32(100.0)).astype(F32)
        tot = (ws + wn).astype(F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            new_t = ((ts * ws + sdf * wn) / tot).astype(F32)
        ts[inl] = new_t[inl]
        ws[inl] = np.clip(tot, F32(0), F32(1))[inl]
