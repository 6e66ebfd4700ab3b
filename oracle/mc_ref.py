"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's CUDA
marching cubes (tools/marching_cubes/marching_cubes.cu:164-424) and of the Python post-processing
around it (utils/pytorch3d_extras.py:90-100).

Parity pin: (1) the triangle table is read from tests/golden/mc_case_tris.npy, observed from the
reference's compiled CPU marching cubes (oracle/derive_mc_table.py); (2) on volumes where the CUDA
and CPU semantics coincide (every cell active, no corner < -0.99999, no degenerate triangle) the
output is compared as a triangle set with oracle/_ref (the compiled reference) in
tests/test_oracle_mc.py.  The CUDA-only rules (active list, bounds, unobserved-corner skip) are
restated from the cited lines; they cannot be executed here (no NVIDIA GPU).
"""
from __future__ import annotations

import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLE = np.load(os.path.join(_HERE, "..", "tests", "golden", "mc_case_tris.npy"))

CODE_TO_VI = [0, 1, 4, 5, 3, 2, 7, 6]  # marching_cubes.cu:176 indexTable
EDGE_CODES = [(0, 1), (1, 5), (4, 5), (0, 4), (2, 3), (3, 7), (6, 7), (2, 6), (0, 2), (1, 3), (5, 7), (4, 6)]  # :312-325
EPS = np.float32(1e-5)
F32 = np.float32


def _interp(iso, p1, p2, v1, v2):
    """vertexInterp, marching_cubes.cu:70-90."""
    if abs(iso - v1) < EPS:
        return p1
    if abs(iso - v2) < EPS:
        return p2
    if abs(v1 - v2) < EPS:
        return p1
    r = F32(iso - v1) / F32(v2 - v1)
    return (p1 * (F32(1) - r) + p2 * r).astype(F32)


def marching_cubes_active(vol, keys, iso=0.0, mn=None, mx=None):
    """vol [X,Y,Z] float32; keys [N,3] (i,j,k) in processing order.  Returns raw
    (verts [V,3] in (x,y,z) = (k,j,i) order, faces [V/3,3], ids [V]) like marching_cubes_."""
    X, Y, Z = vol.shape
    iso = F32(iso)
    W, H, D = Z, Y, X
    hash_mul = W + W * H + W * H * D
    verts, ids = [], []
    for (i, j, k) in np.asarray(keys).tolist():
        if not (0 <= i < X - 1 and 0 <= j < Y - 1 and 0 <= k < Z - 1):
            continue
        if mn is not None and (i < mn[0] or j < mn[1] or k < mn[2]):
            continue
        if mx is not None and (i >= mx[0] or j >= mx[1] or k >= mx[2]):
            continue
        val = np.empty(8, dtype=F32)
        case = 0
        for c in range(8):
            dx, dy, dz = c & 1, (c >> 1) & 1, (c >> 2) & 1
            v = vol[i + dz, j + dy, k + dx]
            val[c] = v
            if v < iso:
                case |= 1 << CODE_TO_VI[c]
        if (val < F32(-0.99999)).any():
            continue
        for e in _TABLE[case]:
            if e == 255:
                break
            c1, c2 = EDGE_CODES[e]
            p1 = np.array([k + (c1 & 1), j + ((c1 >> 1) & 1), i + ((c1 >> 2) & 1)], dtype=F32)
            p2 = np.array([k + (c2 & 1), j + ((c2 >> 1) & 1), i + ((c2 >> 2) & 1)], dtype=F32)
            verts.append(_interp(iso, p1, p2, val[c1], val[c2]))
            v1 = int(p1[0]) + int(p1[1]) * W + int(p1[2]) * W * H
            v2 = int(p2[0]) + int(p2[1]) * W + int(p2[2]) * W * H
            ids.append(v1 * hash_mul + v2)
    V = len(verts)
    verts = np.array(verts, dtype=F32).reshape(V, 3)
    faces = np.arange(V, dtype=np.int64).reshape(-1, 3)
    return verts, faces, np.array(ids, dtype=np.int64)


def postprocess(verts, faces, ids):
    """utils/pytorch3d_extras.py:90-100: dedup by sorted unique id, verts[:, [2,1,0]], faces.flip(1)."""
    if len(verts) == 0:
        return verts, faces
    uniq, inv = np.unique(ids, return_inverse=True)
    v = np.zeros((len(uniq), 3), dtype=F32)
    v[inv] = verts
    return v[:, [2, 1, 0]], inv[faces][:, ::-1]


def triangle_set(verts, faces, decimals=4):
    """Order-independent representation: each triangle as a rotation-normalised tuple of rounded vertex coords."""
    out = []
    for f in faces:
        tri = [tuple(np.round(verts[i], decimals).tolist()) for i in f]
        m = min(range(3), key=lambda q: tri[q])
        out.append(tuple(tri[m:] + tri[:m]))
    return sorted(out)
