"""marching_cubes() with the reference wrapper's signature and post-processing
(reference utils/pytorch3d_extras.py:39-107), on the HIP kernels of csrc/mc.hip.

The reference JIT-builds a CUDA/C++ torch extension at import (:9-17) exposing
``marching_cubes_(vol, isolevel, active_voxels, min_bounds, max_bounds)``; here the native side is
the C ABI pair dt_mc_count / dt_mc_generate and the active set is the fuser's voxel bitmap
(an ``active_voxels`` key list is accepted too and converted).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from .. import _abi


def keys_to_bitmap(active_keys: torch.Tensor, dims) -> torch.Tensor:
    """[N,3] integer voxel keys (i,j,k) -> int32 bitmap words (bit id&31 of word id>>5)."""
    X, Y, Z = dims
    dev = active_keys.device
    k = active_keys.long()
    ids = (k[:, 0] * Y + k[:, 1]) * Z + k[:, 2]
    ok = (k[:, 0] >= 0) & (k[:, 0] < X) & (k[:, 1] >= 0) & (k[:, 1] < Y) & (k[:, 2] >= 0) & (k[:, 2] < Z)
    ids = torch.unique(ids[ok])
    words = torch.zeros((X * Y * Z) // 32, dtype=torch.int64, device=dev)
    words.index_add_(0, ids >> 5, torch.ones_like(ids) << (ids & 31))
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)
    return words.to(torch.int32)


def bitmap_to_keys(bitmap: torch.Tensor, dims) -> torch.Tensor:
    """Inverse of keys_to_bitmap: sorted [N,3] int32 keys."""
    X, Y, Z = dims
    w = bitmap.long() & 0xFFFFFFFF
    nz = torch.nonzero(w).flatten()
    bits = (w[nz, None] >> torch.arange(32, device=bitmap.device)[None]) & 1
    sel = torch.nonzero(bits)
    ids = nz[sel[:, 0]] * 32 + sel[:, 1]
    return torch.stack([ids // (Y * Z), (ids // Z) % Y, ids % Z], 1).to(torch.int32)


def marching_cubes_raw(values_f16: torch.Tensor, bitmap: torch.Tensor, isolevel: float, min_bounds=None, max_bounds=None):
    """The native call: returns (verts [V,3] f32 in (k,j,i) order, faces [V/3,3] i64, ids [V] i64)."""
    if not values_f16.is_cuda:
        raise _abi.DoubletakeHipError("marching cubes only runs on a ROCm GPU (no CPU fallback)")
    L = _abi.lib()
    X, Y, Z = values_f16.shape
    dev = values_f16.device
    vol = values_f16.contiguous()
    if vol.dtype != torch.float16:
        vol = vol.half()
    stream = _abi.current_stream(dev)
    ws = torch.empty(int(L.dt_mc_workspace_bytes(X, Y, Z)) // 4, dtype=torch.int32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)

    def ibuf(b):
        if b is None:
            return None
        vals = [int(v) for v in (b.tolist() if isinstance(b, torch.Tensor) else b)]
        return (C.c_int * 3)(*vals)

    mn, mx = ibuf(min_bounds), ibuf(max_bounds)
    _abi.check(L.dt_mc_count(_abi.ptr(vol), _abi.ptr(bitmap), X, Y, Z, float(isolevel), mn, mx, _abi.ptr(ws),
                             _abi.ptr(counts), stream), "dt_mc_count")
    ncells, nverts = counts.tolist()  # the one host read
    if nverts < 0:
        raise _abi.DoubletakeHipError("marching cubes: vertex count overflows int32")
    verts = torch.empty(nverts, 3, dtype=torch.float32, device=dev)
    faces = torch.empty(nverts // 3, 3, dtype=torch.int64, device=dev)
    ids = torch.empty(nverts, dtype=torch.int64, device=dev)
    _abi.check(L.dt_mc_generate(_abi.ptr(vol), _abi.ptr(bitmap), X, Y, Z, float(isolevel), mn, mx, _abi.ptr(ws),
                                _abi.ptr(verts), _abi.ptr(faces), _abi.ptr(ids), nverts, stream), "dt_mc_generate")
    return verts, faces, ids


def merge_by_edge_id(soup_verts: torch.Tensor, soup_faces: torch.Tensor, edge_ids: torch.Tensor):
    """Triangle soup -> indexed mesh.  Every vertex of the soup lies on a lattice edge and carries that edge's int64 id,
    so vertices with equal ids are the same point: keep one per id (ids ascending) and re-index the faces.
    (What the reference does with torch.unique for its CUDA path, utils/pytorch3d_extras.py:90-96.)"""
    ids_sorted, soup_to_merged = torch.unique(edge_ids, return_inverse=True)
    merged = soup_verts.new_zeros((ids_sorted.numel(), soup_verts.shape[1]))
    merged[soup_to_merged] = soup_verts  # duplicates write identical values
    return merged, soup_to_merged[soup_faces]


def marching_cubes(
    vol_batch: torch.Tensor,
    active_voxels: torch.Tensor,
    isolevel: Optional[float] = None,
    return_local_coords: bool = True,
    min_bounds: Optional[torch.Tensor] = None,
    max_bounds: Optional[torch.Tensor] = None,
) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Reference signature (utils/pytorch3d_extras.py:39-107).  ``active_voxels`` is either the
    [N,3] key list the reference passes or an int32 bitmap (see keys_to_bitmap).  Returns per batch element the
    merged vertices in (x, y, z) order and faces with reversed winding, or empty lists for an empty surface."""
    out_verts, out_faces = [], []
    nx, ny, nz = vol_batch.shape[1:]
    for vol in vol_batch:
        level = float((vol.max() + vol.min()) / 2) if isolevel is None else isolevel
        bitmap = active_voxels if active_voxels.dim() == 1 else keys_to_bitmap(active_voxels, (nx, ny, nz))
        soup, tris, ids = marching_cubes_raw(vol, bitmap, level, min_bounds, max_bounds)
        if soup.shape[0] == 0 or tris.shape[0] == 0:
            out_verts.append([])
            out_faces.append([])
            continue
        if return_local_coords:  # [-1, 1] over the (k, j, i) lattice extents
            half_extent = (vol.new_tensor([nz, ny, nx], dtype=torch.float32) - 1) * 0.5
            soup = soup / half_extent[None] - 1.0
        merged, tris = merge_by_edge_id(soup, tris, ids)
        out_verts.append(merged.flip(1))   # kernel order (k, j, i) -> (i, j, k) = world (x, y, z)
        out_faces.append(tris.flip(1))     # the axis flip mirrors the mesh: reverse the winding with it
    return out_verts, out_faces
