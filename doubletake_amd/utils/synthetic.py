"""Closed-form synthetic inputs for the DoubleTake hot path.

Everything here is produced by an integer hash of the flat element index and a
seed, so the same arrays come out on every platform and numpy/torch version
(no RNG streams).  Used by the golden-vector script, the parity tests, the
smoke test and ``bench.py`` (SURVEY.md section 8(d) "Synthetic inputs").

Shapes follow the reference's tensor-name suffixes:
    cur_feats_bchw, src_feats_bkchw, *_b44 / *_bk44 matrices, hint maps _b1hw.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def hash_u01(shape, seed: int) -> np.ndarray:
    """Uniform [0,1) float32 with 24 random bits per element (splitmix64 finaliser)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) * _M1 + np.uint64(seed & 0xFFFFFFFF) * _M2 + np.uint64(0x1234567)
        x ^= x >> np.uint64(30)
        x *= _M2
        x ^= x >> np.uint64(27)
        x *= _M3
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def hash_normalish(shape, seed: int) -> np.ndarray:
    """Zero-mean, unit-variance float32 (sum of three uniforms, Irwin-Hall)."""
    s = hash_u01(shape, seed) + hash_u01(shape, seed + 7919) + hash_u01(shape, seed + 15485863)
    return ((s - np.float32(1.5)) * np.float32(2.0)).astype(np.float32)


def feature_map(shape_chw_prefix, h, w, seed):
    """Unit-variance features: smooth sinusoidal field + per-texel hash noise.

    shape_chw_prefix: leading dims (e.g. (b, c) or (b, k, c)).
    """
    lead = tuple(shape_chw_prefix)
    nlead = int(np.prod(lead))
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    idx = np.arange(nlead, dtype=np.float64).reshape(-1, 1, 1)
    fx = 0.05 + 0.013 * ((idx * 7 + seed) % 11)
    fy = 0.04 + 0.017 * ((idx * 5 + 3 * seed) % 7)
    ph = 0.37 * idx + 0.11 * seed
    smooth = np.sin(fx * xx[None] + ph) * np.cos(fy * yy[None] - 0.5 * ph)
    smooth = (smooth * np.sqrt(2.0)).astype(np.float32)  # ~unit variance
    noise = hash_normalish((nlead, h, w), seed * 31 + 5)
    out = np.float32(0.7071) * smooth + np.float32(0.7071) * noise
    return out.reshape(lead + (h, w)).astype(np.float32)


def _rodrigues(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * (Kx @ Kx)


def intrinsics(h, w, batch=1):
    """ScanNet-like pinhole K (4x4) for an (h, w) grid and its inverse, float32."""
    K = np.eye(4, dtype=np.float64)
    K[0, 0] = 0.9 * w
    K[1, 1] = 0.9 * w
    K[0, 2] = w / 2.0
    K[1, 2] = h / 2.0
    invK = np.linalg.inv(K)
    K = np.broadcast_to(K, (batch, 4, 4)).astype(np.float32).copy()
    invK = np.broadcast_to(invK, (batch, 4, 4)).astype(np.float32).copy()
    return K, invK


def relative_poses(batch, num_src, seed, behind_view=False):
    """Keyframe-like relative poses.

    Returns (src_extrinsics_bk44 = src_cam_T_cur_cam, src_poses_bk44 = cur_cam_T_src_cam).
    Baselines 0.1-0.3 m on a circle plus +-0.1 m along z, rotations <= 10 degrees.
    With behind_view=True the last source view of every batch element looks
    backwards (exercises the z' <= 0 mask).
    """
    ext = np.zeros((batch, num_src, 4, 4), dtype=np.float64)
    pose = np.zeros_like(ext)
    u = hash_u01((batch, num_src, 8), seed * 101 + 17).astype(np.float64)
    for b in range(batch):
        for k in range(num_src):
            r = u[b, k]
            rad = 0.1 + 0.2 * r[0]
            phi = 2 * np.pi * (k + r[1]) / max(num_src, 1)
            t = np.array([rad * np.cos(phi), rad * np.sin(phi), 0.2 * (r[2] - 0.5)])
            axis = np.array([r[3] - 0.5, r[4] - 0.5, r[5] - 0.5]) + 1e-3
            ang = np.deg2rad(10.0) * r[6]
            R = _rodrigues(axis, ang)
            if behind_view and k == num_src - 1:
                R = _rodrigues([0.0, 1.0, 0.0], np.pi) @ R
            T = np.eye(4)
            T[:3, :3] = R
            T[:3, 3] = t
            pose[b, k] = T  # cur_cam_T_src_cam
            ext[b, k] = np.linalg.inv(T)  # src_cam_T_cur_cam
    return ext.astype(np.float32), pose.astype(np.float32)


def hint_maps(batch, H2, W2, seed, empty=False, hole_frac=0.3):
    """depth_hint_b1hw (NaN where invalid), depth_hint_mask_b1hw, sampled_weights_b1hw."""
    if empty:
        d = np.full((batch, 1, H2, W2), np.nan, dtype=np.float32)
        m = np.zeros((batch, 1, H2, W2), dtype=np.float32)
        wgt = np.zeros((batch, 1, H2, W2), dtype=np.float32)
        return d, m, wgt
    yy, xx = np.meshgrid(np.arange(H2, dtype=np.float64), np.arange(W2, dtype=np.float64), indexing="ij")
    base = 1.5 + 1.5 * hash_u01((batch,), seed + 3).astype(np.float64).reshape(batch, 1, 1, 1)
    relief = 0.25 * np.sin(0.031 * xx + 0.2 * seed) * np.cos(0.027 * yy) + 0.002 * (xx - W2 / 2)
    d = (base + relief[None, None]).astype(np.float32)
    holes = hash_u01((batch, 1, H2, W2), seed * 13 + 1) < np.float32(hole_frac)
    m = (~holes).astype(np.float32)
    d = np.where(holes, np.float32(np.nan), d).astype(np.float32)
    wgt = hash_u01((batch, 1, H2, W2), seed * 17 + 2)
    return d, m, wgt


def volume_inputs(batch, num_src, h, w, channels=16, seed=0, empty_hint=False, behind_view=False):
    """All inputs of the cost-volume forward call as a dict of float32 numpy arrays."""
    K, invK = intrinsics(h, w, batch)
    ext, pose = relative_poses(batch, num_src, seed, behind_view=behind_view)
    dh, dm, dw = hint_maps(batch, 2 * h, 2 * w, seed, empty=empty_hint)
    return {
        "cur_feats": feature_map((batch, channels), h, w, seed * 3 + 1),
        "src_feats": feature_map((batch, num_src, channels), h, w, seed * 3 + 2),
        "src_extrinsics": ext,
        "src_poses": pose,
        "src_Ks": np.broadcast_to(K[:, None], (batch, num_src, 4, 4)).copy(),
        "cur_invK": invK,
        "min_depth": np.full((batch, 1, 1, 1), 0.25, dtype=np.float32),
        "max_depth": np.full((batch, 1, 1, 1), 5.0, dtype=np.float32),
        "depth_hint_b1hw": dh,
        "depth_hint_mask_b1hw": dm,
        "sampled_weights_b1hw": dw,
    }


def formula_weights(shape, seed, scale=None):
    """Deterministic 'random' weights ~ U(-a, a) with a = scale or 1/sqrt(fan_in)."""
    shape = tuple(shape)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    a = np.float32(scale if scale is not None else 1.0 / np.sqrt(max(fan_in, 1)))
    return ((hash_u01(shape, seed) * np.float32(2.0) - np.float32(1.0)) * a).astype(np.float32)


def prior_pyramid(batch, widths, h0, w0, seed):
    """Image-prior feature pyramid: level i has widths[i] channels at (h0 >> i, w0 >> i)."""
    return [hash_normalish((batch, c, h0 >> i, w0 >> i), seed + 100 * i) for i, c in enumerate(widths)]


def tsdf_frames(num_frames, H=480, W=640, seed=0, bounds=None):
    """Depth maps at 1.5-3 m, full-res intrinsics and a camera path inside the bounds.

    Returns depth_b1hw f32, K_b44 f32, cam_T_world_b44 f32 (world -> camera extrinsics).
    """
    if bounds is None:
        bounds = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depths, Ks, Ts = [], [], []
    fx = 577.0 * W / 640.0
    K = np.eye(4)
    K[0, 0] = fx
    K[1, 1] = fx
    K[0, 2] = (W - 1) / 2.0
    K[1, 2] = (H - 1) / 2.0
    cx = 0.5 * (bounds["xmin"] + bounds["xmax"])
    cy = 0.5 * (bounds["ymin"] + bounds["ymax"])
    cz = 0.5 * (bounds["zmin"] + bounds["zmax"])
    for f in range(num_frames):
        u = hash_u01((8,), seed * 1009 + f).astype(np.float64)
        d = 2.25 + 0.75 * np.sin(0.011 * xx + 0.7 * f) * np.cos(0.013 * yy + 0.3 * seed)
        d = d + 0.02 * (hash_u01((H, W), seed * 7 + f * 3 + 1).astype(np.float64) - 0.5)
        depths.append(d.astype(np.float32))
        yaw = 2 * np.pi * (f / max(num_frames, 1)) + 0.3 * u[0]
        pitch = np.deg2rad(20.0) * (u[1] - 0.5)
        # camera looks along +z of its own frame; world z is up.
        R_wc = _rodrigues([0, 0, 1], yaw) @ _rodrigues([1, 0, 0], -np.pi / 2 + pitch)
        t_wc = np.array([cx + 0.8 * (u[2] - 0.5), cy + 0.8 * (u[3] - 0.5), cz + 0.4 * (u[4] - 0.5)])
        Twc = np.eye(4)
        Twc[:3, :3] = R_wc
        Twc[:3, 3] = t_wc
        Ts.append(np.linalg.inv(Twc))
        Ks.append(K.copy())
    return (
        np.stack(depths)[:, None].astype(np.float32),
        np.stack(Ks).astype(np.float32),
        np.stack(Ts).astype(np.float32),
    )


def formula_params(shapes, seed, scale_mult=1.0):
    """Deterministic parameters for a list of shapes in ``named_parameters()`` order.

    Same rule as tests/golden/make_golden.py:set_formula_weights -- parameter j uses seed
    + 1000*j; matrices/filters ~ U(-a, a) with a = scale_mult*sqrt(3/fan_in), vectors
    (biases) ~ U(-0.1, 0.1).
    """
    out = []
    for j, shp in enumerate(shapes):
        shp = tuple(int(s) for s in shp)
        if len(shp) > 1:
            a = scale_mult * np.sqrt(3.0 / int(np.prod(shp[1:])))
        else:
            a = 0.1
        out.append(formula_weights(shp, seed + 1000 * j, scale=a))
    return out


def mlp_param_shapes(channels):
    """Shapes of nn.Linear weights/biases for an MLP with the given channel list."""
    shapes = []
    for i in range(len(channels) - 1):
        shapes += [(channels[i + 1], channels[i]), (channels[i + 1],)]
    return shapes


def mlp_in_channels(num_src, feat_dim=16):
    """Matching-MLP input width (reference modules/mesh_hint_volume.py:49-67)."""
    return feat_dim * (1 + num_src) + (1 + num_src) + 3 * (1 + num_src) + num_src * 3 + 3 * num_src
