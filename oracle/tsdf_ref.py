"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's fp16
TSDF volume, fuser and sampler (tools/tsdf.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity pin: tests/golden/tsdf.npz = the reference's own TSDF / TSDFFuser / sample_tsdf run on CPU
in half precision in the build container (tests/golden/make_golden.py:gen_tsdf); checked
bit-for-bit (index sets, half values, half weights) in tests/test_oracle_tsdf.py.  The
reference's production path is the same code on a CUDA device; bit-parity with *that* cannot be
established without an NVIDIA GPU, so the pin is the CPU-half run.

Rounding model (fitted to the golden run, torch 2.10 CPU): every torch op on half tensors
computes in float32 and rounds its result to half once; Python-scalar operands of binary ops
enter at float32 precision ("original_scalar_value"), comparisons against Python scalars are made
after rounding the scalar to half; half matmuls accumulate in float32; grid_sample on half
tensors runs torch's scalar fallback in which every intermediate is rounded to half.
"""
from __future__ import annotations

import numpy as np

F16 = np.float16
F32 = np.float32
VOX_MOD = 8


def h(x):
    """round to half (returns float16 array)."""
    return np.asarray(x, dtype=F32).astype(F16)


def f(x):
    return np.asarray(x).astype(F32)


def volume_dims(bounds, voxel_size):
    """TSDF.from_bounds, tools/tsdf.py:134-142: ceil((max-min)/vs/8)*8 per axis (Python floats)."""
    d = []
    for a in "xyz":
        d.append(int(np.ceil((bounds[a + "max"] - bounds[a + "min"]) / voxel_size / VOX_MOD)) * VOX_MOD)
    return tuple(d)


def voxel_coords(bounds, voxel_size):
    """TSDF.generate_voxel_coords (:157-166) then .half() (:146-148): fp32 origin + idx*vs -> half.
    Returns [3, X, Y, Z] float16 and the float32 origin."""
    dims = volume_dims(bounds, voxel_size)
    origin = np.array([bounds["xmin"], bounds["ymin"], bounds["zmin"]], dtype=F32)
    grid = np.stack(np.meshgrid(*[np.arange(n, dtype=np.int64) for n in dims], indexing="ij"), 0)
    # torch: LongTensor * python float -> float32 tensor (scalar at fp32), then + float32 origin
    coords = origin.reshape(3, 1, 1, 1) + grid.astype(F32) * F32(voxel_size)
    return coords.astype(F16), origin


class TSDFVolume:
    """State of reference class TSDF (:53-84): fp16 values (-1), weights (0), coords, active set."""

    def __init__(self, bounds, voxel_size):
        self.voxel_size = float(voxel_size)
        self.coords, origin32 = voxel_coords(bounds, voxel_size)
        self.origin = origin32.astype(F16)  # TSDF.__init__: origin.half() (:76)
        self.values = np.full(self.coords.shape[1:], -1.0, dtype=F16)
        self.weights = np.zeros(self.coords.shape[1:], dtype=F16)
        self.active = set()

    @property
    def dims(self):
        return self.values.shape


def _inv_half(M16):
    """torch.inverse(M.float()).half() (:452-453)."""
    return np.linalg.inv(f(M16)).astype(F32).astype(F16)


def _matmul_half(A16, B16):
    """half @ half with float32 accumulation, one rounding to half."""
    return (f(A16) @ f(B16)).astype(F32).astype(F16)


def frustum_bounds(invK16, pose16, min_depth, max_depth, img_h, img_w):
    """get_frustum_bounds (:15-50) in half."""
    corners = np.array([[0, 0, 1, 1], [img_w, 0, 1, 1], [0, img_h, 1, 1], [img_w, img_h, 1, 1]], dtype=F32).astype(F16).T
    cp = _matmul_half(invK16, corners)  # [4,4]
    mn = cp.copy()
    mn[:3] = h(f(mn[:3]) * F32(min_depth))
    mx = cp.copy()
    mx[:3] = h(f(mx[:3]) * F32(max_depth))
    c8 = np.concatenate([mn, mx], axis=1)  # [4,8]
    c8 = _matmul_half(pose16, c8)
    return c8.min(axis=1)[:3], c8.max(axis=1)[:3]


def _grid_sample_nearest_half(depth16, gx16, gy16):
    """F.grid_sample(half input, half grid, nearest, zeros, align_corners=False) on CPU (:480-486):
    torch's scalar fallback templated on Half -- ((g + 1) * size - 1) / 2 with every intermediate
    rounded to half, then nearbyint, zeros out of bounds."""
    H, W = depth16.shape

    def unnorm(g16, size):
        t = h(f(g16) + F32(1.0))
        t = h(f(t) * F32(size))
        t = h(f(t) - F32(1.0))
        return h(f(t) / F32(2.0))

    ix = unnorm(gx16, W)
    iy = unnorm(gy16, H)
    with np.errstate(invalid="ignore"):
        xn = np.rint(f(ix))
        yn = np.rint(f(iy))
    ok = np.isfinite(xn) & np.isfinite(yn) & (xn >= 0) & (xn < W) & (yn >= 0) & (yn < H)
    xi = np.where(ok, xn, 0).astype(np.int64)
    yi = np.where(ok, yn, 0).astype(np.int64)
    return np.where(ok, depth16[yi, xi], F16(0.0)).astype(F16)


def integrate(vol: TSDFVolume, depth_hw, K44, cam_T_world44, max_depth, min_depth=0.5, extended_neg_truncation=False):
    """TSDFFuser.integrate_depth for one frame (:444-558).  depth/K/T are cast to half first as
    OurFuser.fuse_frames does (tools/fusers_helper.py:67-73).  Returns (valid_linear_ids,
    active_keys[n,3]) of this frame (sorted) and updates vol in place."""
    depth16 = np.asarray(depth_hw, dtype=F32).astype(F16)
    K16 = np.asarray(K44, dtype=F32).astype(F16)
    T16 = np.asarray(cam_T_world44, dtype=F32).astype(F16)
    H, W = depth16.shape
    vs = vol.voxel_size
    trunc = 3.0 * vs  # truncation_size * voxel_size (:362,397-399), Python double
    depth_max = max_depth + trunc + 0.1
    invK16 = _inv_half(K16)
    pose16 = _inv_half(T16)
    bmin, bmax = frustum_bounds(invK16, pose16, 0.01, depth_max, H, W)
    c = vol.coords
    inside = np.ones(c.shape[1:], dtype=bool)
    for a in range(3):
        inside &= (c[a] > bmin[a]) & (c[a] < bmax[a])  # strict (:459-466)
    ids = np.flatnonzero(inside.reshape(-1))
    X = np.stack([c[0].reshape(-1)[ids], c[1].reshape(-1)[ids], c[2].reshape(-1)[ids], np.ones(ids.size, dtype=F16)], 0)
    P = _matmul_half(K16, T16)[:3]  # (:407)
    q = _matmul_half(P, X)  # [3,N] (:409)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = h(f(q[0]) / f(q[2]))
        v = h(f(q[1]) / f(q[2]))  # (:410)
        # 2 * pix / img_size - 1 (:477), img_size is a half tensor (W, H)
        gx = h(f(h(f(h(F32(2.0) * f(u))) / f(F16(W)))) - F32(1.0))
        gy = h(f(h(f(h(F32(2.0) * f(v))) / f(F16(H)))) - F32(1.0))
    sd = _grid_sample_nearest_half(depth16, gx, gy)
    # Voxels whose half pixel coordinates overflow to +-inf (|z| ~ 1e-3, i.e. in the camera plane):
    # ATen's CPU grid sampler converts inf to an integer, which is C++ undefined behaviour -- the
    # imported reference returns a *random pixel* for about half of them.  The CUDA sampler
    # returns 0 (out of bounds), which is what this oracle (and the HIP kernel) do.  They are
    # reported so the golden comparison can skip them.
    vol.last_undefined_ids = ids[~(np.isfinite(f(gx)) & np.isfinite(f(gy)))]
    vd = q[2]
    # confidence (:490-497)
    t = h(f(sd) - F32(min_depth))
    t = h(f(t) / F32(max_depth - min_depth))
    t = h(F32(1.0) - f(t))
    t = np.clip(t, F16(0.25), F16(1.0))
    conf = h(f(t) * f(t))
    with np.errstate(invalid="ignore"):
        dist = h(f(sd) - f(vd))  # (:500)
        tsdf = np.clip(h(f(dist) / F32(trunc)), F16(-1.0), F16(1.0))  # (:501)
        trunc_check = -trunc * 1.5 if extended_neg_truncation else -trunc
        valid = (vd > F16(0)) & (dist > F16(trunc_check)) & (sd > F16(0)) & (vd < F16(max_depth)) & (conf > F16(0))
        active = valid & (dist < F16(trunc))  # (:530)
    vids = ids[valid]
    old_v = vol.values.reshape(-1)[vids]
    old_w = vol.weights.reshape(-1)[vids]
    new_t = tsdf[valid]
    cf = conf[valid]
    new_w = h(f(h(f(cf) * F32(2.5))) / F32(100.0))  # (:546-549)
    tot = h(f(old_w) + f(new_w))
    num = h(f(h(f(old_v) * f(old_w))) + f(h(f(new_t) * f(new_w))))
    vol.values.reshape(-1)[vids] = h(f(num) / f(tot))  # (:553-555)
    vol.weights.reshape(-1)[vids] = np.minimum(tot, F16(1.0))  # (:556-558)
    aids = ids[active]
    Xd, Yd, Zd = vol.dims
    keys = np.stack([aids // (Yd * Zd), (aids // Zd) % Yd, aids % Zd], 1).astype(np.int64)
    vol.active.update(map(tuple, keys.tolist()))
    return np.sort(vids), keys


def sample(vol: TSDFVolume, points_N3, what="weights", fp16_math=False):
    """TSDF.sample_tsdf (:277-339): trilinear, align_corners=True, zeros padding.  On CPU the
    reference casts volume and grid to fp32 (:327-330); fp16_math=True mimics its GPU branch."""
    pts = np.asarray(points_N3, dtype=F32)
    origin = f(vol.origin).reshape(1, 3)  # origin is stored in half (:76), promoted by the subtraction
    vc = (pts - origin).astype(F32)
    vc = (vc / F32(vol.voxel_size)).astype(F32)
    dims = np.array(vol.dims, dtype=F32).reshape(1, 3)
    vc = (vc / (dims - F32(1.0))).astype(F32)
    vc = (vc * F32(2.0) - F32(1.0)).astype(F32)
    src = f(vol.weights if what == "weights" else vol.values)
    if fp16_math:
        vc = f(h(vc))
    # align_corners=True unnormalise: ((g + 1) / 2) * (size - 1)
    idx = ((vc + F32(1.0)) / F32(2.0)) * (dims - F32(1.0))
    if fp16_math:
        idx = f(h(idx))
    out = np.zeros(pts.shape[0], dtype=F32)
    i0 = np.floor(idx)
    fr = (idx - i0).astype(F32)
    i0 = i0.astype(np.int64)
    X, Y, Z = vol.dims
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                xi, yi, zi = i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz
                ok = (xi >= 0) & (xi < X) & (yi >= 0) & (yi < Y) & (zi >= 0) & (zi < Z)
                wx = fr[:, 0] if dx else F32(1.0) - fr[:, 0]
                wy = fr[:, 1] if dy else F32(1.0) - fr[:, 1]
                wz = fr[:, 2] if dz else F32(1.0) - fr[:, 2]
                val = src[np.where(ok, xi, 0), np.where(ok, yi, 0), np.where(ok, zi, 0)]
                out += np.where(ok, val * (wx * wy * wz), F32(0.0)).astype(F32)
    return out.astype(F16) if fp16_math else out
