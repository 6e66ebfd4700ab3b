"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's
plane-sweep cost volume in numpy float32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (doubletake_amd/) never does.

Parity pin: checked against golden vectors captured from the imported reference
(tests/golden/volume_*.npz, made by tests/golden/make_golden.py) in
tests/test_oracle_volume.py.

Every function cites the reference lines it restates (paths relative to
/root/reference/src/doubletake/).  The arithmetic that the reference delegates to torch
(F.grid_sample, nn.Linear, F.normalize, F.cosine_similarity; torch pinned at 2.0.1 in
environment.yml:13) is restated from torch's documented semantics.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def generate_depth_planes(min_depth, max_depth, num_bins):
    """modules/cost_volume.py:96-130 (ramp buffer :60-61).  Returns [b, D] float32."""
    ramp = np.linspace(0, 1, num_bins, dtype=F32).reshape(1, num_bins)
    mn = np.asarray(min_depth, dtype=F32).reshape(-1, 1)
    mx = np.asarray(max_depth, dtype=F32).reshape(-1, 1)
    logp = np.log(mn) + np.log(mx / mn) * ramp
    return np.exp(logp).astype(F32)


def pixel_centres(h, w):
    """utils/geometry_utils.py:34-45: (x+0.5, y+0.5, 1), row-major N = y*w + x.  [3, N]."""
    xx, yy = np.meshgrid(np.arange(w, dtype=F32), np.arange(h, dtype=F32), indexing="xy")
    pix = np.stack([xx + F32(0.5), yy + F32(0.5), np.ones_like(xx)], 0)
    return pix.reshape(3, -1).astype(F32)


def backproject(depth_bN, invK_b44, h, w):
    """utils/geometry_utils.py:55-63 BackprojectDepth.forward -> [b, 4, N]."""
    pix = pixel_centres(h, w)
    cam = np.matmul(invK_b44[:, :3, :3].astype(F32), pix[None])  # [b,3,N]
    cam = depth_bN[:, None, :].astype(F32) * cam
    ones = np.ones_like(cam[:, :1])
    return np.concatenate([cam, ones], 1).astype(F32)


def project(points_B4N, K_B44, T_B44, eps=1e-8):
    """utils/geometry_utils.py:77-93 Project3D.forward -> [B, 3, N] = (u, v, z + eps)."""
    P = np.matmul(K_B44.astype(F32), T_B44.astype(F32))
    cam = np.matmul(P[:, :3], points_B4N)
    z = cam[:, 2:3]
    mask = np.abs(z) > F32(eps)
    depth = (z + F32(eps)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(mask, F32(1.0) / depth, F32(1.0)).astype(F32)
    pix = cam[:, :2] * scale
    return np.concatenate([pix, depth], 1).astype(F32)


def grid_sample_bilinear_zeros(src_Bchw, u_BN, v_BN, h, w):
    """F.grid_sample(mode=bilinear, padding_mode=zeros, align_corners=False) as called at
    modules/cost_volume.py:188-196 with grid = 2*uv*(1/w,1/h) - 1 (:186).

    Follows torch's unnormalisation ((g+1)*size - 1)/2, taps at floor/floor+1, out-of-bounds
    taps contribute zero.  Returns [B, c, N].
    """
    B, c = src_Bchw.shape[:2]
    gx = F32(2.0) * u_BN * F32(1.0 / w) - F32(1.0)
    gy = F32(2.0) * v_BN * F32(1.0 / h) - F32(1.0)
    ix = ((gx + F32(1.0)) * F32(w) - F32(1.0)) / F32(2.0)
    iy = ((gy + F32(1.0)) * F32(h) - F32(1.0)) / F32(2.0)
    with np.errstate(invalid="ignore"):
        x0f = np.floor(ix)
        y0f = np.floor(iy)
    wx1 = (ix - x0f).astype(F32)
    wy1 = (iy - y0f).astype(F32)
    wx0 = (F32(1.0) - wx1).astype(F32)
    wy0 = (F32(1.0) - wy1).astype(F32)
    out = np.zeros((B, c, u_BN.shape[1]), dtype=F32)
    flat = src_Bchw.reshape(B, c, h * w)
    big = 1 << 30
    # non-finite coordinates sample nothing (all taps out of bounds)
    finite = np.isfinite(ix) & np.isfinite(iy) & (np.abs(ix) < 1e9) & (np.abs(iy) < 1e9)
    x0 = np.where(finite, x0f, -big).astype(np.int64)
    y0 = np.where(finite, y0f, -big).astype(np.int64)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = x0 + dx
            yi = y0 + dy
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            idx = np.where(ok, yi * w + xi, 0)
            vals = np.take_along_axis(flat, np.broadcast_to(idx[:, None, :], (B, c, idx.shape[1])), axis=2)
            wgt = np.where(ok, wx * wy, F32(0.0)).astype(F32)
            wgt = np.where(np.isfinite(wgt), wgt, F32(0.0))
            out += vals * wgt[:, None, :]
    return out


def warp_features(src_feats_bkchw, src_ext_bk44, src_Ks_bk44, cur_invK_b44, plane_b):
    """modules/cost_volume.py:132-217 for a single depth plane: plane_b is [b] (one depth per batch element, what
    generate_depth_planes yields) or [b, h*w] (a slice of a caller-supplied depth_planes_bdhw, :249-250).

    Returns world_points [b*k,4,N], depths [b,k,N] (z'), warped [b,k,c,N], mask [b,k,N],
    pix [b,k,2,N].
    """
    b, k, c, h, w = src_feats_bkchw.shape
    N = h * w
    depth_bN = np.broadcast_to(np.asarray(plane_b, dtype=F32).reshape(b, -1), (b, N))
    world_b4N = backproject(depth_bN, cur_invK_b44, h, w)
    world_B4N = np.repeat(world_b4N, k, axis=0)
    cam = project(world_B4N, src_Ks_bk44.reshape(-1, 4, 4), src_ext_bk44.reshape(-1, 4, 4))
    u, v, z = cam[:, 0], cam[:, 1], cam[:, 2]
    warped = grid_sample_bilinear_zeros(src_feats_bkchw.reshape(b * k, c, h, w), u, v, h, w)
    mask = (z > 0).astype(F32)
    return (
        world_B4N,
        z.reshape(b, k, N),
        warped.reshape(b, k, c, N),
        mask.reshape(b, k, N),
        cam[:, :2].reshape(b, k, 2, N),
    )


def _plane_list(min_depth, max_depth, num_bins, planes_bdhw, b, N):
    """planes [b, D] from generate_depth_planes, or the caller's depth_planes_bdhw flattened to [b, D, N]
    (modules/cost_volume.py:249-250: `if depth_planes_bdhw is None: ... generate_depth_planes`)."""
    if planes_bdhw is None:
        return generate_depth_planes(min_depth, max_depth, num_bins)
    return np.asarray(planes_bdhw, dtype=F32).reshape(b, num_bins, N)


def dot_cost_volume(cur_feats, src_feats, src_ext, src_Ks, cur_invK, min_depth, max_depth, num_bins, planes_bdhw=None):
    """CostVolumeManager.build_cost_volume, modules/cost_volume.py:219-315 -> [b, D, h, w]."""
    b, k, c, h, w = src_feats.shape
    planes = _plane_list(min_depth, max_depth, num_bins, planes_bdhw, b, h * w)
    cur = cur_feats.reshape(b, 1, c, h * w)
    out = np.zeros((b, num_bins, h * w), dtype=F32)
    for d in range(num_bins):
        _, _, warped, mask, _ = warp_features(src_feats, src_ext, src_Ks, cur_invK, planes[:, d])
        dot = (warped * cur).sum(axis=2, dtype=F32) * mask
        out[:, d] = dot.sum(axis=1, dtype=F32)
    return out.reshape(b, num_bins, h, w), planes


def lowest_cost(volume_bdhw, planes_bd):
    """modules/cost_volume.py:317-320,355-361: plane depth at argmax over d (first max); planes_bd is [b, D] or
    [b, D, h*w] / [b, D, h, w] (per-pixel planes: the gather of indices_to_disparity)."""
    idx = np.argmax(volume_bdhw, axis=1)
    b, D, h, w = volume_bdhw.shape
    planes = np.broadcast_to(np.asarray(planes_bd, dtype=F32).reshape(b, D, -1), (b, D, h * w)).reshape(b, D, h, w)
    return np.take_along_axis(planes, idx[:, None], axis=1)[:, 0].astype(F32)


def pose_distance(pose_B44):
    """utils/geometry_utils.py:187-199 -> (combined, R_measure, t_measure), each [B]."""
    R = pose_B44[:, :3, :3].astype(F32)
    t = pose_B44[:, :3, 3].astype(F32)
    tr = (R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]).astype(F32)
    Rm = np.sqrt(F32(2.0) * (F32(1.0) - np.minimum(F32(3.0), tr) / F32(3.0))).astype(F32)
    tm = np.sqrt((t * t).sum(1, dtype=F32)).astype(F32)
    return np.sqrt(tm * tm + Rm * Rm).astype(F32), Rm, tm


def _normalize(x, axis, eps=1e-12):
    """F.normalize(dim): x / max(||x||, eps)."""
    n = np.sqrt((x * x).sum(axis=axis, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(eps))).astype(F32)


def leaky_relu(x, slope):
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def mlp_forward(x_Nc, weights, slope=0.01):
    """modules/networks.py:120-135 MLP with disable_final_activation=True.

    weights = [(W0, b0), (W1, b1), ...] with W of shape [out, in] (nn.Linear layout);
    LeakyReLU default slope 0.01 between layers, none after the last.
    """
    y = x_Nc.astype(F32)
    for i, (W, bias) in enumerate(weights):
        y = (y @ W.T.astype(F32) + bias.astype(F32)).astype(F32)
        if i + 1 < len(weights):
            y = leaky_relu(y, slope)
    return y


def get_mask(pix_bk2N, h, w):
    """modules/cost_volume.py:73-94: 2 < u < w-2 and 2 < v < h-2."""
    u = pix_bk2N[:, :, 0]
    v = pix_bk2N[:, :, 1]
    return (u > 2) & (u < w - 2) & (v > 2) & (v < h - 2)


def nearest_resize(x_b1HW, h, w):
    """F.interpolate(mode='nearest') as used at modules/mesh_hint_volume.py:186-202:
    src index = floor(dst * (in/out)) clipped to in-1."""
    H, W = x_b1HW.shape[-2:]
    ys = np.minimum(np.floor(np.arange(h, dtype=F32) * F32(H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w, dtype=F32) * F32(W / w)).astype(np.int64), W - 1)
    return x_b1HW[..., ys[:, None], xs[None, :]]


def mlp_input_features(cur_feats, src_feats, src_ext, src_poses, src_Ks, cur_invK, plane_b):
    """MLP input assembly for ONE plane, modules/mesh_hint_volume.py:209-370
    (identical in modules/feature_volume.py).  Returns feats [b, N, Cin], z', mask, pix.

    Channel order (:353-370): warped src feats (k-major), cur feats, mask_k, z'_k, plane,
    dot_k*mask_k, ray-angle_k, cur ray xyz, src rays (k, xyz), pose dist_k, R_k, t_k.
    """
    b, k, c, h, w = src_feats.shape
    N = h * w
    world_B4N, z, warped, mask, pix = warp_features(src_feats, src_ext, src_Ks, cur_invK, plane_b)
    X = world_B4N[:, :3].reshape(b, k, 3, N)
    cur_rays = _normalize(X, axis=2)  # :290-299
    t_src = src_poses[:, :, :3, 3].astype(F32)  # get_camera_rays, utils/geometry_utils.py:177-182
    src_rays = _normalize(X - t_src[:, :, :, None], axis=2)  # :303-310
    # F.cosine_similarity(eps=1e-5) (:329-331)
    w12 = (cur_rays * src_rays).sum(2, dtype=F32)
    n1 = np.maximum(np.sqrt((cur_rays * cur_rays).sum(2, dtype=F32)), F32(1e-5))
    n2 = np.maximum(np.sqrt((src_rays * src_rays).sum(2, dtype=F32)), F32(1e-5))
    angle = (w12 / (n1 * n2)).astype(F32)
    cur = cur_feats.reshape(b, 1, c, N)
    dot = ((warped * cur).sum(2, dtype=F32) * mask).astype(F32)  # :334-340
    pd, Rm, tm = pose_distance(src_poses.reshape(-1, 4, 4))  # :153-175
    ones = np.ones((1, 1, N), dtype=F32)
    plane_map = np.broadcast_to(np.asarray(plane_b, dtype=F32).reshape(b, 1, -1), (b, 1, N))
    feats = np.concatenate(
        [
            warped.reshape(b, k * c, N),
            cur_feats.reshape(b, c, N),
            mask,
            z,
            plane_map,
            dot,
            angle,
            cur_rays[:, 0],
            src_rays.reshape(b, k * 3, N),
            pd.reshape(b, k, 1) * ones,
            Rm.reshape(b, k, 1) * ones,
            tm.reshape(b, k, 1) * ones,
        ],
        axis=1,
    ).astype(F32)
    return feats.transpose(0, 2, 1), z, mask, pix


def feature_volume(
    cur_feats, src_feats, src_ext, src_poses, src_Ks, cur_invK, min_depth, max_depth, num_bins, mlp_weights,
    hint=None, hint_mlp_weights=None, return_mask=None, planes_bdhw=None,
):
    """FeatureVolumeManager.build_cost_volume (modules/feature_volume.py:81-356) when
    hint is None, FeatureMeshHintVolumeManager.build_cost_volume
    (modules/mesh_hint_volume.py:84-393) otherwise.

    hint = dict(depth_hint_b1hw, sampled_weights_b1hw, depth_hint_mask_b1hw) at (H2, W2).
    return_mask: None | "slow" (per-view mask of the LAST plane, :270-287) | "fast"
    (any_k depth AND any_k bounds at the last plane, :818-822).
    planes_bdhw: the optional depth_planes_bdhw argument (:95,149-150), any [b,D,h,w].
    Returns (volume [b,D,h,w], planes [b,D] -- [b,D,h*w] when planes_bdhw is given --, mask or None).
    """
    b, k, c, h, w = src_feats.shape
    N = h * w
    planes = _plane_list(min_depth, max_depth, num_bins, planes_bdhw, b, N)
    if hint is not None:
        hd = nearest_resize(hint["depth_hint_b1hw"], h, w).reshape(b, N)
        hw_ = nearest_resize(hint["sampled_weights_b1hw"], h, w).reshape(b, N).astype(F32).copy()
        hm = nearest_resize(hint["depth_hint_mask_b1hw"], h, w).reshape(b, N) != 0
        hw_[~hm] = 0  # :204
    out = np.zeros((b, num_bins, N), dtype=F32)
    mask_out = None
    for d in range(num_bins):
        feats, z, mask, pix = mlp_input_features(cur_feats, src_feats, src_ext, src_poses, src_Ks, cur_invK, planes[:, d])
        s = mlp_forward(feats.reshape(b * N, -1), mlp_weights).reshape(b, N)
        if hint is not None:
            with np.errstate(invalid="ignore"):
                hmap = np.abs(hd - planes[:, d].reshape(b, -1)).astype(F32)  # :213
            hmap = np.where(hm, hmap, F32(-1.0)).astype(F32)  # :214
            hin = np.stack([s, hmap, hw_], -1).reshape(b * N, 3)
            s = mlp_forward(hin, hint_mlp_weights).reshape(b, N)  # :373-386
        out[:, d] = s
        if d == num_bins - 1 and return_mask:
            dm = z > 0
            bm = get_mask(pix, h, w)
            if return_mask == "slow":
                mask_out = (dm & bm).reshape(b, k, h, w)
            else:
                mask_out = (dm.any(1) & bm.any(1)).reshape(b, h, w)
    return out.reshape(b, num_bins, h, w), planes, mask_out
