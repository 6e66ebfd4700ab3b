"""OurFuser with the reference's constructor and methods (reference tools/fusers_helper.py:23-107)."""
from __future__ import annotations

import numpy as np

from .tsdf import TSDF, TSDFFuser


class _Verts:
    def __init__(self, v):
        self.vertices = v


def _ply_vertices(path):
    """Minimal PLY vertex reader (ascii / binary_little_endian) -- only the bounds are needed."""
    with open(path, "rb") as f:
        fmt, nverts, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element"):
                in_vertex = line.split()[1] == "vertex"
                if in_vertex:
                    nverts = int(line.split()[2])
            elif line.startswith("property") and in_vertex:
                props.append((line.split()[-1], line.split()[1]))
            elif line == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=nverts, ndmin=2)
            names = [p[0] for p in props]
            return np.stack([data[:, names.index(a)] for a in "xyz"], 1)
        tmap = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
                "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "short": "i2", "ushort": "u2"}
        dt = np.dtype([(n, "<" + tmap[t]) for n, t in props])
        arr = np.frombuffer(f.read(nverts * dt.itemsize), dtype=dt, count=nverts)
        return np.stack([arr[a].astype(np.float64) for a in "xyz"], 1)


class DepthFuser:
    def __init__(self, gt_path="", fusion_resolution=0.04, max_fusion_depth=3.0, fuse_color=False):
        self.fusion_resolution = fusion_resolution
        self.max_fusion_depth = max_fusion_depth


class OurFuser(DepthFuser):
    def __init__(self, gt_path="", fusion_resolution=0.04, max_fusion_depth=3, fuse_color=False,
                 extended_neg_truncation=False, bounds=None):
        """``bounds`` (dict xmin..zmax) is an extension for callers that know the extent without a mesh."""
        super().__init__(gt_path, fusion_resolution, max_fusion_depth, fuse_color)
        if bounds is not None:
            tsdf_pred = TSDF.from_bounds(bounds, voxel_size=fusion_resolution)
        elif gt_path is not None and gt_path != "":
            tsdf_pred = TSDF.from_mesh(_Verts(_ply_vertices(gt_path)), voxel_size=fusion_resolution)
        else:
            b = {"xmin": -10.0, "xmax": 10.0, "ymin": -10.0, "ymax": 10.0, "zmin": -10.0, "zmax": 10.0}
            tsdf_pred = TSDF.from_bounds(b, voxel_size=fusion_resolution)
        self.extended_neg_truncation = extended_neg_truncation
        self.tsdf_fuser_pred = TSDFFuser(tsdf_pred, max_depth=max_fusion_depth)

    def fuse_frames(self, depths_b1hw, K_b44, cam_T_world_b44, color_b3hw=None):
        self.tsdf_fuser_pred.integrate_depth(
            depth_b1hw=depths_b1hw.half(), cam_T_world_T_b44=cam_T_world_b44.half(), K_b44=K_b44.half(),
            extended_neg_truncation=self.extended_neg_truncation)

    def export_mesh(self, path, export_single_mesh=True, trim_tsdf_using_confience=False):
        _, verts, faces = self.get_mesh_pytorch3d()
        from ..utils.formats import write_ply

        write_ply(path, verts.cpu().numpy(), faces.cpu().numpy())

    def save_tsdf(self, path):
        self.tsdf_fuser_pred.tsdf.save_tsdf(path)

    def sample_tsdf(self, world_points_N3, what_to_sample="tsdf", sampling_method="bilinear"):
        return self.tsdf_fuser_pred.tsdf.sample_tsdf(world_points_N3, what_to_sample=what_to_sample,
                                                     sampling_method=sampling_method)

    def get_mesh(self, export_single_mesh=True, convert_to_trimesh=True):
        return self.tsdf_fuser_pred.tsdf.to_mesh(export_single_mesh=export_single_mesh)

    def get_mesh_pytorch3d(self, scale_to_world=True, min_bounds_3=None, max_bounds_3=None):
        return self.tsdf_fuser_pred.tsdf.to_mesh_pytorch3d(scale_to_world=scale_to_world, min_bounds_3=min_bounds_3,
                                                           max_bounds_3=max_bounds_3)
