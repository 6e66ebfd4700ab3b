"""ORACLE (test infrastructure, NOT product code) -- torch-CPU restatement of the hot path, the
"reference-equivalent CPU path" of BASELINE.md section 3 that bench.py times as ``cpu_baseline``
(kind "port": the reference's Python cannot travel to the GPU box).

It composes the same ATen operators the reference composes -- F.grid_sample, F.normalize,
F.cosine_similarity, torch.cat of the per-plane MLP input, nn.functional.linear, F.conv2d,
F.interpolate -- in the same per-plane loop (``hint_volume_loop`` = the reference's slow manager,
modules/mesh_hint_volume.py:84-393) or as one batched pass over all planes (``hint_volume_batched`` =
its Fast manager, :679-928), so that its CPU time is what the reference's would be on the same cores.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity pin: tests/test_oracle_torch_cpu.py checks every function against the golden vectors captured
from the imported reference (tests/golden/volume_*.npz, networks.npz).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def depth_planes(min_depth, max_depth, D, b):
    """modules/cost_volume.py:96-130 -> [b, D]."""
    ramp = torch.linspace(0, 1, D).view(1, D)
    lo = min_depth.reshape(-1, 1).float()
    hi = max_depth.reshape(-1, 1).float()
    return torch.exp(torch.log(lo) + torch.log(hi / lo) * ramp).expand(b, D)


def _pixel_grid(h, w):
    """utils/geometry_utils.py:34-45 -> [3, h*w]: (x+0.5, y+0.5, 1)."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    return torch.stack([xs.reshape(-1) + 0.5, ys.reshape(-1) + 0.5, torch.ones(h * w)], 0)


def pose_metrics(poses_B44):
    """utils/geometry_utils.py:187-199."""
    R = poses_B44[:, :3, :3]
    t = poses_B44[:, :3, 3]
    tr = R.diagonal(dim1=-2, dim2=-1).sum(-1)
    Rm = torch.sqrt(2 * (1 - torch.clamp(tr, max=3.0) / 3))
    tm = t.norm(dim=1)
    return torch.sqrt(tm ** 2 + Rm ** 2), Rm, tm


def _mlp(x, wts, slope=0.01):
    """modules/networks.py:120-135 (LeakyReLU 0.01 between layers, none after the last)."""
    for i, (W, b) in enumerate(wts):
        x = F.linear(x, W, b)
        if i + 1 < len(wts):
            x = F.leaky_relu(x, slope)
    return x


def _warp(src_Bchw, P_B34, rays_b3N, plane_b, k, h, w):
    """Back-project at one plane depth, project into every source view, bilinear fetch
    (modules/cost_volume.py:132-217).  plane_b: [b] or [b,1]."""
    b = rays_b3N.shape[0]
    X = rays_b3N * plane_b.view(b, 1, 1)                                   # [b,3,N]
    X4 = torch.cat([X, torch.ones_like(X[:, :1])], 1).repeat_interleave(k, dim=0)  # [b*k,4,N]
    cam = P_B34 @ X4
    z = cam[:, 2:3]
    zz = z + 1e-8
    scale = torch.where(z.abs() > 1e-8, 1.0 / zz, torch.ones_like(zz))
    uv = cam[:, :2] * scale
    grid = uv.permute(0, 2, 1).reshape(b * k, h, w, 2)
    norm = 2 * grid * torch.tensor([1.0 / w, 1.0 / h]).view(1, 1, 1, 2) - 1
    warped = F.grid_sample(src_Bchw, norm, mode="bilinear", padding_mode="zeros", align_corners=False)
    return X, uv, zz, warped


def _plane_features(cur, src_B, P, rays, plane_b, tsrc_bk3, pose_feats, b, k, c, h, w):
    """The 16(K+1)+(K+1)+3(K+1)+6K-channel MLP input of one plane, channels-last
    (modules/mesh_hint_volume.py:216-370)."""
    X, uv, zz, warped = _warp(src_B, P, rays, plane_b, k, h, w)
    warped = warped.view(b, k, c, h, w)
    z = zz.view(b, k, h, w)
    mask = (z > 0).float()
    Xk = X.view(b, 1, 3, h, w).expand(b, k, 3, h, w)
    cur_ray = F.normalize(Xk, dim=2)
    src_ray = F.normalize(Xk - tsrc_bk3.view(b, k, 3, 1, 1), dim=2)
    angle = F.cosine_similarity(cur_ray, src_ray, dim=2, eps=1e-5)
    dot = (warped * cur.unsqueeze(1)).sum(2) * mask
    plane_map = plane_b.view(b, 1, 1, 1).expand(b, 1, h, w)
    feats = torch.cat([warped.reshape(b, k * c, h, w), cur, mask, z, plane_map, dot, angle, cur_ray[:, 0],
                       src_ray.reshape(b, k * 3, h, w), *pose_feats], 1)
    return feats.permute(0, 2, 3, 1), uv.view(b, k, 2, h, w), z


def _hint_maps(hint, h, w):
    """modules/mesh_hint_volume.py:186-204."""
    hd = F.interpolate(hint["depth_hint_b1hw"], size=(h, w), mode="nearest")
    hw = F.interpolate(hint["sampled_weights_b1hw"], size=(h, w), mode="nearest").clone()
    hm = F.interpolate(hint["depth_hint_mask_b1hw"], size=(h, w), mode="nearest").bool()
    hw[~hm] = 0
    return hd, hw, hm


def _common(cur, src, src_ext, src_poses, src_Ks, cur_invK):
    b, k, c, h, w = src.shape
    P = (src_Ks.reshape(-1, 4, 4) @ src_ext.reshape(-1, 4, 4))[:, :3]
    rays = cur_invK[:, :3, :3] @ _pixel_grid(h, w).unsqueeze(0)
    pd, Rm, tm = pose_metrics(src_poses.reshape(-1, 4, 4))
    pose_feats = [v.view(b, k, 1, 1).expand(b, k, h, w) for v in (pd, Rm, tm)]
    return b, k, c, h, w, P, rays, src_poses[:, :, :3, 3], pose_feats


@torch.no_grad()
def hint_volume_loop(cur, src, src_ext, src_poses, src_Ks, cur_invK, min_depth, max_depth, D, mlp, hint=None, hint_mlp=None,
                     plane_ids=None):
    """Loop over planes (reference slow manager).  hint=None -> FeatureVolumeManager
    (modules/feature_volume.py:81-356).  Returns (volume [b,D,h,w], planes [b,D]).
    plane_ids: evaluate only these planes of the D (bench.py's bounded CPU-baseline sample: every plane costs the same ops);
    the returned volume then has len(plane_ids) planes."""
    b, k, c, h, w, P, rays, tsrc, pose_feats = _common(cur, src, src_ext, src_poses, src_Ks, cur_invK)
    planes = depth_planes(min_depth, max_depth, D, b)
    src_B = src.reshape(b * k, c, h, w)
    if hint is not None:
        hd, hw, hm = _hint_maps(hint, h, w)
    out = []
    for d in (range(D) if plane_ids is None else plane_ids):
        feats, _, _ = _plane_features(cur, src_B, P, rays, planes[:, d], tsrc, pose_feats, b, k, c, h, w)
        s = _mlp(feats, mlp)                                                 # [b,h,w,1]
        if hint is not None:
            hmap = (hd - planes[:, d].view(b, 1, 1, 1)).abs()
            hmap[~hm] = -1
            s = _mlp(torch.cat([s, hmap.permute(0, 2, 3, 1), hw.permute(0, 2, 3, 1)], -1), hint_mlp)
        out.append(s.squeeze(-1))
    return torch.stack(out, 1), planes


@torch.no_grad()
def hint_volume_batched(cur, src, src_ext, src_poses, src_Ks, cur_invK, min_depth, max_depth, D, mlp, hint=None,
                        hint_mlp=None):
    """All planes in one pass by folding D into the batch (reference Fast manager, modules/mesh_hint_volume.py:679-928)."""
    b, k, c, h, w, P, rays, tsrc, pose_feats = _common(cur, src, src_ext, src_poses, src_Ks, cur_invK)
    planes = depth_planes(min_depth, max_depth, D, b)
    rep = lambda t: t.repeat_interleave(D, dim=0)
    feats, _, _ = _plane_features(rep(cur), rep(src).reshape(b * D * k, c, h, w), rep(P.view(b, k, 3, 4)).reshape(-1, 3, 4),
                                  rep(rays), planes.reshape(-1), rep(tsrc), [rep(p) for p in pose_feats], b * D, k, c, h, w)
    s = _mlp(feats, mlp)
    if hint is not None:
        hd, hw, hm = _hint_maps(hint, h, w)
        hmap = (rep(hd) - planes.reshape(-1, 1, 1, 1)).abs()
        hmap[~rep(hm)] = -1
        s = _mlp(torch.cat([s, hmap.permute(0, 2, 3, 1), rep(hw).permute(0, 2, 3, 1)], -1), hint_mlp)
    return s.view(b, D, h, w), planes


@torch.no_grad()
def dot_volume(cur, src, src_ext, src_Ks, cur_invK, min_depth, max_depth, D):
    """CostVolumeManager.build_cost_volume (modules/cost_volume.py:219-315), loop over planes."""
    b, k, c, h, w = src.shape
    P = (src_Ks.reshape(-1, 4, 4) @ src_ext.reshape(-1, 4, 4))[:, :3]
    rays = cur_invK[:, :3, :3] @ _pixel_grid(h, w).unsqueeze(0)
    planes = depth_planes(min_depth, max_depth, D, b)
    src_B = src.reshape(b * k, c, h, w)
    out = []
    for d in range(D):
        _, _, zz, warped = _warp(src_B, P, rays, planes[:, d], k, h, w)
        mask = (zz.view(b, k, h, w) > 0).float()
        out.append(((warped.view(b, k, c, h, w) * cur.unsqueeze(1)).sum(2) * mask).sum(1))
    return torch.stack(out, 1), planes


def lowest_cost(volume, planes):
    """modules/cost_volume.py:317-320,355-361."""
    idx = volume.argmax(1, keepdim=True)
    return torch.gather(planes.view(*planes.shape, 1, 1).expand_as(volume), 1, idx)[:, 0]


# ---- conv stacks (modules/layers.py:33-94, networks.py:88-117,20-85, networks_fast.py:6-141) ----------------
def _sub(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def _conv(x, p, name, stride=1):
    W = p[name + ".weight"]
    return F.conv2d(x, W, p.get(name + ".bias"), stride=stride, padding=W.shape[-1] // 2)


def basic_block(x, p, stride=1):
    out = F.leaky_relu(_conv(x, p, "conv1", stride), 0.2)
    out = _conv(out, p, "conv2")
    idt = _conv(x, p, "downsample.0", stride) if "downsample.0.weight" in p else x
    return F.leaky_relu(out + idt, 0.2)


@torch.no_grad()
def cv_encoder(x, img_feats, p):
    outs = []
    for i, f in enumerate(img_feats):
        x = basic_block(x, _sub(p, f"convs.ds_conv_{i}."), stride=1 if i == 0 else 2)
        x = torch.cat([x, f], 1)
        x = basic_block(x, _sub(p, f"convs.conv_{i}.0."))
        x = basic_block(x, _sub(p, f"convs.conv_{i}.1."))
        outs.append(x)
    return outs


def _conv_block(x, p):
    return F.elu(_conv(F.elu(_conv(x, p, "conv1")), p, "conv2"))


@torch.no_grad()
def skip_decoder_regression(features, p):
    out = {}
    x = features[-1]
    for bi, scale in ((1, 3), (2, 2), (3, 1), (4, 0)):
        q = _sub(p, f"block{bi}.")
        x = _conv_block(x, _sub(q, "pre_concat_conv."))
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _conv_block(torch.cat([x, features[-1 - bi]], 1), _sub(q, "post_concat_conv."))
        out[f"feature_s{scale}_b1hw"] = x
        hp = _sub(p, f"out{bi}.")
        y = F.elu(_conv(x, hp, "0"))
        y = F.elu(_conv(y, hp, "2"))
        out[f"log_depth_pred_s{scale}_b1hw"] = _conv(y, hp, "4")
    return out


def _up(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


@torch.no_grad()
def depth_decoder_pp(input_features, p, nodes=None):
    """nodes: optional dict that receives every UNet++ node output X_ij under its ModuleDict name ``in_conv_{i}{j}`` (what a
    forward hook on the reference's ``convs[name]`` sees)."""
    prev = list(input_features)
    outputs, depth = [], {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            ins = [basic_block(prev[i], _sub(p, f"convs.right_conv_{i}{j - 1}.")),
                   _up(basic_block(prev[i + 1], _sub(p, f"convs.diag_conv_{i + 1}{j - 1}.")))]
            if i + j != 4:
                ins.append(_up(basic_block(outputs[-1], _sub(p, f"convs.up_conv_{i + 1}{j}."))))
            q = _sub(p, f"convs.in_conv_{i}{j}.")
            out = basic_block(basic_block(torch.cat(ins, 1), _sub(q, "0.")), _sub(q, "conv_0."))
            outputs.append(out)
            if nodes is not None:
                nodes[f"in_conv_{i}{j}"] = out
            hp = _sub(p, f"convs.output_{i}.")
            y = basic_block(out, _sub(hp, "0.")) if i != 0 else out
            depth[f"log_depth_pred_s{i}_b1hw"] = _conv(y, hp, "1")
        prev = outputs[::-1]
    return depth
