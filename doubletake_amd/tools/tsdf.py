"""fp16 TSDF volume + fuser with the reference's class/method names (reference tools/tsdf.py),
on the HIP kernels of csrc/tsdf.hip and csrc/mc.hip.

Differences by design:
  * the [3,X,Y,Z] fp16 coordinate volume is not stored -- kernels recompute
    half(fp32(origin) + idx*voxel_size) from the voxel index (``voxel_coords_3hwd`` materialises
    it on demand for API users);
  * the open3d HashSet of active keys is a device bitmap (``voxel_bitmap``; ``active_keys()``
    lists it in ascending voxel order);
  * GPU-resident only (no CPU fallback): ``cuda()`` / ``cpu()`` are accepted and ignored.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Tuple

import numpy as np
import torch

from .. import _abi
from ..utils.pytorch3d_extras import bitmap_to_keys, marching_cubes


class Meshes:
    """Minimal stand-in for pytorch3d.structures.Meshes (not installed here): verts/faces lists."""

    def __init__(self, verts, faces, textures=None):
        self._verts, self._faces, self.textures = list(verts), list(faces), textures

    def verts_list(self):
        return self._verts

    def faces_list(self):
        return self._faces

    def verts_packed(self):
        return self._verts[0]

    def faces_packed(self):
        return self._faces[0]


def get_frustum_bounds(invK_144, world_T_cam_144, min_depth=0.1, max_depth=10.0, img_h=480, img_w=640):
    """Axis-aligned world bounds of a camera frustum between two depths (reference tools/tsdf.py:15-50) ->
    (minbounds_3, maxbounds_3).  Host-level helper in the caller's dtype/device; the integrate kernel computes the
    same box itself (dt_tsdf_frames_setup_f16)."""
    uv = invK_144.new_tensor([[0, 0, 1, 1], [img_w, 0, 1, 1], [0, img_h, 1, 1], [img_w, img_h, 1, 1]]).t().unsqueeze(0)
    rays = torch.matmul(invK_144, uv)  # [1,4,4]: one column per image corner
    near, far = rays.clone(), rays.clone()
    near[:, :3] *= min_depth
    far[:, :3] *= max_depth
    corners = torch.matmul(world_T_cam_144, torch.cat((near, far), dim=2))[0]  # [4,8]
    return corners.amin(dim=1)[:3], corners.amax(dim=1)[:3]


def _device():
    if not torch.cuda.is_available():
        raise _abi.DoubletakeHipError("doubletake_amd TSDF needs a ROCm GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _recover_origin_f32(voxel_coords_3hwd, voxel_size):
    """A saved volume only holds the half-rounded coordinate grid (reference tools/tsdf.py:267-275), while
    the kernels regenerate coordinates as half(origin_f32 + idx * voxel_size).  Find an fp32 origin that
    reproduces the stored grid bit for bit: every stored half value bounds origin + idx*voxel_size to its
    rounding interval; intersect the intervals per axis, pick a point, verify by regeneration."""
    if voxel_coords_3hwd is None:
        raise NotImplementedError("a TSDF needs either origin_f32 or the voxel_coords_3hwd grid")
    c = voxel_coords_3hwd.half().cpu().numpy()
    vs = np.float32(voxel_size)
    origin = np.zeros(3, dtype=np.float32)
    for a in range(3):
        line = np.moveaxis(c[a], a, 0).reshape(c.shape[1 + a], -1)
        if not (line == line[:, :1]).all():
            raise NotImplementedError("voxel_coords_3hwd is not an axis-aligned regular grid")
        h = line[:, 0]
        step = (np.arange(len(h), dtype=np.float32) * vs).astype(np.float64)  # torch: int64 index * python float -> fp32
        up = np.nextafter(h, np.float16(np.inf)).astype(np.float64)
        dn = np.nextafter(h, np.float16(-np.inf)).astype(np.float64)
        hv = h.astype(np.float64)
        lo = ((hv + dn) * 0.5 - step).max()
        hi = ((hv + up) * 0.5 - step).min()
        found = None
        for frac in (0.5, 0.25, 0.75, 0.125, 0.875, 0.375, 0.625, 0.0, 1.0):
            cand = np.float32(lo + (hi - lo) * frac)
            regen = (cand + np.arange(len(h), dtype=np.float32) * vs).astype(np.float16)
            if lo <= hi and np.array_equal(regen, h):
                found = cand
                break
        if found is None:
            raise NotImplementedError("voxel_coords_3hwd is not half(origin + idx*voxel_size) for any fp32 origin")
        origin[a] = found
    return origin


#: Arithmetic of the trilinear sampling of the half volume (TSDF.sample_tsdf and the fused hint kernel) when the caller
#: does not say.  False: fp32 blend of the half voxels -- the reference's CPU branch (tools/tsdf.py:327-330), which is what
#: the goldens pin.  True: grid and result rounded to half and the 0.025 cut compared in half, as the reference does when the
#: volume lives on the GPU; that branch cannot be executed here (no CUDA device), so it stays an opt-in model of it.  What the
#: choice changes on the incremental loop is measured by tests/test_incremental_cfg4_gpu.py (hint pixels within half
#: precision of the cut) and stated in DESIGN.md section 2.  Environment override: DT_TSDF_SAMPLE_FP16=1.
SAMPLE_FP16_MATH = os.environ.get("DT_TSDF_SAMPLE_FP16", "0") == "1"


class TSDF:
    VOX_MOD = 8

    def __init__(self, voxel_coords_3hwd, tsdf_values, tsdf_weights, voxel_size, origin, origin_f32=None, device=None):
        """Reference signature (tools/tsdf.py:62-84) + the fp32 origin the coordinates were generated
        from (TSDF.from_bounds passes it; when absent it is recovered from voxel_coords_3hwd)."""
        dev = device or _device()
        self.device = dev
        self.voxel_size = float(voxel_size)
        self.tsdf_values = tsdf_values.to(device=dev, dtype=torch.float16).contiguous()
        self.tsdf_weights = tsdf_weights.to(device=dev, dtype=torch.float16).contiguous()
        self.origin = origin.to(dtype=torch.float16).cpu()
        X, Y, Z = self.tsdf_values.shape
        if origin_f32 is None:
            # the half origin is not enough to regenerate the coordinates bit-for-bit; check it
            o = self.origin.float().numpy()
            regen = self.generate_voxel_coords(torch.from_numpy(o), (X, Y, Z), self.voxel_size).half()
            if voxel_coords_3hwd is not None and torch.equal(regen, voxel_coords_3hwd.half().cpu()):
                origin_f32 = o
            else:
                origin_f32 = _recover_origin_f32(voxel_coords_3hwd, self.voxel_size)
        self.origin_f32 = np.asarray(origin_f32, dtype=np.float32).reshape(3)
        self.voxel_bitmap = torch.zeros((X * Y * Z) // 32, dtype=torch.int32, device=dev)
        self._mc_ws = None

    # -- constructors ---------------------------------------------------------------------------
    @classmethod
    def from_file(cls, tsdf_file):
        data = np.load(tsdf_file)
        return cls(torch.from_numpy(data["voxel_coords_3hwd"]), torch.from_numpy(data["tsdf_values"]),
                   torch.from_numpy(data["tsdf_weights"]), data["voxel_size"].item(), torch.from_numpy(data["origin"]))

    @classmethod
    def from_mesh(cls, mesh, voxel_size: float):
        """mesh: anything with a .vertices [N,3] array (tools/tsdf.py:99-120: bounds padded by 3 voxels)."""
        v = np.asarray(mesh.vertices)
        xmax, ymax, zmax = v.max(0)
        xmin, ymin, zmin = v.min(0)
        bounds = {"xmin": xmin, "xmax": xmax, "ymin": ymin, "ymax": ymax, "zmin": zmin, "zmax": zmax}
        for key, val in bounds.items():
            bounds[key] = val - 3 * voxel_size if "min" in key else val + 3 * voxel_size
        return cls.from_bounds(bounds, voxel_size)

    @classmethod
    def volume_dims(cls, bounds: dict, voxel_size: float):
        """tools/tsdf.py:134-142 -- integer arithmetic on Python floats, identical on every backend."""
        for key in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax"):
            if key not in bounds:
                raise KeyError("Provided bounds dict need to have keys'xmin', 'xmax', 'ymin', 'ymax', 'zmin', 'zmax'!")
        return tuple(int(np.ceil((bounds[a + "max"] - bounds[a + "min"]) / voxel_size / cls.VOX_MOD)) * cls.VOX_MOD
                     for a in "xyz")

    @classmethod
    def from_bounds(cls, bounds: dict, voxel_size: float, device=None):
        dims = cls.volume_dims(bounds, voxel_size)
        dev = device or _device()
        origin = torch.FloatTensor([bounds["xmin"], bounds["ymin"], bounds["zmin"]])
        values = -torch.ones(dims, dtype=torch.float16, device=dev)
        weights = torch.zeros(dims, dtype=torch.float16, device=dev)
        return cls(None, values, weights, voxel_size, origin, origin_f32=origin.numpy().copy(), device=dev)

    @classmethod
    def generate_voxel_coords(cls, origin: torch.Tensor, volume_dims: Tuple[int, int, int], voxel_size: float):
        grid = torch.meshgrid([torch.arange(vd) for vd in volume_dims], indexing="ij")
        return origin.view(3, 1, 1, 1) + torch.stack(grid, 0) * voxel_size

    @property
    def voxel_coords_3hwd(self):
        c = self.generate_voxel_coords(torch.from_numpy(self.origin_f32), tuple(self.tsdf_values.shape), self.voxel_size)
        return c.half().to(self.device)

    def cuda(self):
        return self

    def cpu(self):
        return self

    # -- active set ---------------------------------------------------------------------------------
    def active_keys(self):
        """[N,3] int32 keys of the active voxels (what the reference keeps in its HashSet)."""
        return bitmap_to_keys(self.voxel_bitmap, tuple(self.tsdf_values.shape))

    # -- meshing --------------------------------------------------------------------------------------
    def to_mesh_pytorch3d(self, scale_to_world=True, min_bounds_3=None, max_bounds_3=None):
        """tools/tsdf.py:216-255 -> (Meshes, verts, faces)."""
        dev = self.device
        org = self.origin.float().to(dev)
        mn = torch.floor((min_bounds_3.to(dev) - org) / self.voxel_size).int() if min_bounds_3 is not None else None
        mx = torch.ceil((max_bounds_3.to(dev) - org) / self.voxel_size).int() if max_bounds_3 is not None else None
        bv, bf = marching_cubes(self.tsdf_values[None], self.voxel_bitmap, isolevel=0.0, return_local_coords=False,
                                min_bounds=mn, max_bounds=mx)
        verts, faces = bv[0], bf[0]
        if len(verts) == 0:
            verts = torch.zeros(1, 3, device=dev)
            faces = torch.zeros(1, 3, device=dev)
        if scale_to_world:
            verts = self.origin.view(1, 3).to(dev) + verts * self.voxel_size
        return Meshes(verts=[verts], faces=[faces]), verts, faces

    def to_mesh(self, scale_to_world=True, export_single_mesh=False):
        raise NotImplementedError(
            "final mesh export goes through a scikit-image fork on CPU in the reference (tools/tsdf.py:182-214); "
            "out of scope (SURVEY.md section 8a M2) -- use to_mesh_pytorch3d()")

    def save_mesh(self, savepath, filename):
        """tools/tsdf.py:257-265 writes ``filename`` (".bin" -> ".ply") under ``savepath``.  The reference meshes with
        a scikit-image fork on the CPU there; this writes the GPU marching-cubes mesh (to_mesh_pytorch3d)."""
        import os

        from ..utils.formats import write_ply

        os.makedirs(savepath, exist_ok=True)
        _, verts, faces = self.to_mesh_pytorch3d(scale_to_world=True)
        write_ply(os.path.join(savepath, filename).replace(".bin", ".ply"), verts.cpu().numpy(), faces.cpu().numpy())

    def save_tsdf(self, filepath):
        """tools/tsdf.py:267-275 (same npz keys)."""
        np.savez_compressed(
            filepath,
            tsdf_values=self.tsdf_values.cpu().numpy().astype(np.float16),
            tsdf_weights=self.tsdf_weights.cpu().numpy().astype(np.float16),
            origin=self.origin.cpu().numpy().astype(np.float16),
            voxel_coords_3hwd=self.voxel_coords_3hwd.cpu().numpy().astype(np.float16),
            voxel_size=self.voxel_size,
        )

    # -- sampling ---------------------------------------------------------------------------------------
    def sample_tsdf(self, world_points_N3, what_to_sample="tsdf", sampling_method="bilinear", fp16_math=None):
        """tools/tsdf.py:277-339 (trilinear, align_corners=True).  fp32 math on the half volume (the
        reference's pinned CPU branch) unless fp16_math=True (None: the module default SAMPLE_FP16_MATH)."""
        if fp16_math is None:
            fp16_math = SAMPLE_FP16_MATH
        if not (world_points_N3.ndim == 2 and world_points_N3.shape[1] == 3):
            raise ValueError("world_points_N3 must have shape (N, 3)! Instead got shape {}".format(world_points_N3.shape))
        if sampling_method not in ("bilinear", "trilinear"):
            raise NotImplementedError("only trilinear sampling is used by the reference drivers")
        if what_to_sample not in ("tsdf", "weights"):
            raise ValueError(what_to_sample)
        L = _abi.lib()
        dev = self.device
        pts = world_points_N3.to(device=dev, dtype=torch.float32).contiguous()
        vol = self.tsdf_values if what_to_sample == "tsdf" else self.tsdf_weights
        out = torch.empty(pts.shape[0], dtype=torch.float32, device=dev)
        X, Y, Z = vol.shape
        o = (C.c_float * 3)(*[float(v) for v in self.origin.float().tolist()])
        _abi.check(L.dt_tsdf_sample_f16(_abi.ptr(vol), o, self.voxel_size, X, Y, Z, _abi.ptr(pts), _abi.ptr(out),
                                        pts.shape[0], int(fp16_math), _abi.current_stream(dev)), "dt_tsdf_sample_f16")
        return out.half() if fp16_math else out


class TSDFFuser:
    """Reference tools/tsdf.py:342-558."""

    def __init__(self, tsdf, min_depth=0.5, max_depth=5.0, use_gpu=True):
        if not use_gpu:
            raise _abi.DoubletakeHipError("the doubletake_amd fuser is GPU-only")
        self.tsdf = tsdf
        self.min_depth = min_depth
        self.max_depth = max_depth
        self.use_gpu = True
        self.truncation_size = 3.0
        self.maxW = 100.0
        #: (x_begin, x_end) or None: restrict integrate_depth to this x-slab of the volume (multi-GPU voxel-slab fusion,
        #: parallel.KeyframeShardFuser(mode="slab")); voxels outside are left untouched
        self.x_range = None
        L = _abi.lib()
        self._fp_floats = int(L.dt_tsdf_frame_params_floats())
        self._frame_params = torch.empty(self._fp_floats, dtype=torch.float32, device=tsdf.device)

    voxel_size = property(lambda self: self.tsdf.voxel_size)
    tsdf_values = property(lambda self: self.tsdf.tsdf_values)
    tsdf_weights = property(lambda self: self.tsdf.tsdf_weights)
    shape = property(lambda self: self.tsdf.tsdf_values.shape)
    truncation = property(lambda self: self.truncation_size * self.voxel_size)
    voxel_coords_3hwd = property(lambda self: self.tsdf.voxel_coords_3hwd)
    #: the reference keeps an open3d HashSet of active voxel keys here; this is the same set as an [N,3] int32 tensor
    voxel_hashset = property(lambda self: self.tsdf.active_keys())

    def project_to_camera(self, cam_T_world_T_144, K_144, valid_voxels_14N):
        """tools/tsdf.py:401-412: pixel coordinates and depth of homogeneous world points, [1,3,N] (host-level
        helper in the inputs' dtype; integrate_depth projects inside its kernel)."""
        dev = self.tsdf.device
        P = torch.matmul(K_144.to(dev), cam_T_world_T_144.to(dev))[:, :3]
        cam = torch.matmul(P, valid_voxels_14N.to(dev))
        cam[:, :2] = cam[:, :2] / cam[:, 2, None]
        return cam

    def _thresholds(self, extended_neg_truncation):
        trunc = self.truncation
        th = _abi.TsdfThresholds()
        th.trunc = float(np.float32(trunc))
        th.thr_neg = float(np.float16(-trunc * 1.5 if extended_neg_truncation else -trunc))
        th.thr_pos = float(np.float16(trunc))
        th.max_depth_h = float(np.float16(self.max_depth))
        th.min_depth = float(np.float32(self.min_depth))
        th.depth_range = float(np.float32(self.max_depth - self.min_depth))
        return th

    @torch.no_grad()
    def integrate_depth(self, depth_b1hw, cam_T_world_T_b44, K_b44, depth_mask_b1hw=None, extended_neg_truncation=False):
        """Integrates depth maps into the volume, frame by frame in batch order (tools/tsdf.py:414-558)."""
        L = _abi.lib()
        t = self.tsdf
        dev = t.device
        stream = _abi.current_stream(dev)
        depth = depth_b1hw.to(dev)
        if depth_mask_b1hw is not None:
            depth = depth.clone()
            depth[~depth_mask_b1hw.to(dev)] = -1
        # fp32 depth maps are rounded to half inside the kernel (same values as .half(), no converting copy)
        depth32 = depth.dtype == torch.float32
        depth = depth.contiguous() if depth32 else depth.half().contiguous()
        K16 = K_b44.to(dev).half().contiguous()
        T16 = cam_T_world_T_b44.to(dev).half().contiguous()
        img_h, img_w = depth.shape[2:]
        X, Y, Z = t.tsdf_values.shape
        th = self._thresholds(extended_neg_truncation)
        depth_min = 0.01
        depth_max = self.max_depth + self.truncation + 0.1
        o = (C.c_float * 3)(*[float(v) for v in t.origin_f32])
        nf = depth.shape[0]
        if self._frame_params.numel() < nf * self._fp_floats:
            self._frame_params = torch.empty(nf * self._fp_floats, dtype=torch.float32, device=dev)
        # all frames of the batch in two launches; the kernel applies them in batch order per voxel
        _abi.check(L.dt_tsdf_frames_setup_f16(_abi.ptr(K16), _abi.ptr(T16), nf, img_h, img_w, float(np.float32(depth_min)),
                                              float(np.float32(depth_max)), _abi.ptr(self._frame_params), stream),
                   "dt_tsdf_frames_setup_f16")
        if self.x_range is not None:
            x0, x1 = int(self.x_range[0]), int(self.x_range[1])
            _abi.check(L.dt_tsdf_integrate_frames_xslab_f16(
                _abi.ptr(t.tsdf_values), _abi.ptr(t.tsdf_weights), _abi.ptr(t.voxel_bitmap), o, float(np.float32(t.voxel_size)),
                X, Y, Z, x0, x1 - x0, _abi.ptr(depth), int(depth32), nf, img_h, img_w, _abi.ptr(self._frame_params), C.byref(th),
                stream), "dt_tsdf_integrate_frames_xslab_f16")
            return
        entry = L.dt_tsdf_integrate_frames_f32depth_f16 if depth32 else L.dt_tsdf_integrate_frames_f16
        _abi.check(entry(_abi.ptr(t.tsdf_values), _abi.ptr(t.tsdf_weights), _abi.ptr(t.voxel_bitmap), o,
                         float(np.float32(t.voxel_size)), X, Y, Z, _abi.ptr(depth), nf, img_h, img_w,
                         _abi.ptr(self._frame_params), C.byref(th), stream), "dt_tsdf_integrate_frames_f16")
