"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the hint-mesh depth render.

Reference: utils/rendering_utils.py:9-53 calls PyTorch3D 0.7.4 MeshRasterizer (image_size=(h,w),
blur_radius=0, faces_per_pixel=1) through cameras_from_opencv_projection and keeps fragments.zbuf
(background -1).  PyTorch3D is a third-party dependency absent from /root/reference and from this
image, and the reference has no test or golden for the call site, so it cannot be run: this
file restates the documented semantics -- pixel centres at (x+0.5, y+0.5), a pixel is covered when
its centre is STRICTLY inside the projected triangle (no culling), zbuf = perspective-correct depth of the
nearest covering face, faces with a vertex nearer than 1e-2 dropped -- independently of the kernel's
code path (all pixels per face, vectorised) so that the two can disagree.  Pinned by the hand-derived
known-answer cases of tests/golden/make_raster_handcases.py (PyTorch3D 0.7.4's rules for this call, applied in exact
rational arithmetic, each cited to the file of the release that states it); faces crossing the camera plane are a stated
deviation (dropped here, wrapped around by PyTorch3D).
"""
from __future__ import annotations

import numpy as np


def render_depth(verts_world, faces, cam_T_world, K, h, w):
    V = np.asarray(verts_world, dtype=np.float64)
    T = np.asarray(cam_T_world, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    Xc = V @ T[:3, :3].T + T[:3, 3]
    z = Xc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K[0, 0] * Xc[:, 0] / z + K[0, 2]
        v = K[1, 1] * Xc[:, 1] / z + K[1, 2]
    px, py = np.meshgrid(np.arange(w) + 0.5, np.arange(h) + 0.5)
    depth = np.full((h, w), np.inf)
    for f in np.asarray(faces):
        if (z[f] <= 1e-2).any():
            continue
        (x0, x1, x2), (y0, y1, y2), (z0, z1, z2) = u[f], v[f], z[f]
        area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
        if abs(area) < 1e-12:
            continue
        b0 = ((x1 - px) * (y2 - py) - (x2 - px) * (y1 - py)) / area
        b1 = ((x2 - px) * (y0 - py) - (x0 - px) * (y2 - py)) / area
        b2 = ((x0 - px) * (y1 - py) - (x1 - px) * (y0 - py)) / area
        inside = (b0 > 0) & (b1 > 0) & (b2 > 0)  # strict: PyTorch3D CheckPixelInsideFace (rule R3 of the hand cases)
        with np.errstate(divide="ignore", invalid="ignore"):
            zz = (b0 + b1 + b2) / (b0 / z0 + b1 / z1 + b2 / z2)
        depth = np.where(inside & (zz > 0) & (zz < depth), zz, depth)
    return np.where(np.isfinite(depth), depth, -1.0).astype(np.float32)
