"""GPU: the reference-shaped entry points end to end on one GPU --
  * DepthModelCVHint.forward("test", cur_data, src_data) (doubletake_model.py:265-425) with an image encoder attached and
    the HIP matching encoder, against forward_from_features on the same features;
  * the offline two-pass loop (test_offline_two_pass.py:26-131, :292-500) through parallel.run_two_pass + loops.two_pass_fns
    with world size 1: pass 1 with empty hints fills the hint TSDF, pass 2 renders hints from its mesh and fuses the final
    volume; the hint TSDF must equal fusing pass 1's depths serially, and pass 2 must really see hints."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

BD = dict(xmin=-2.0, xmax=2.0, ymin=-2.0, ymax=2.0, zmin=0.0, zmax=2.4)


class TinyImageEncoder(nn.Module):
    """Stand-in for the timm image-prior network (out of scope): five maps at strides 2..32 with resnet18d's widths."""

    def __init__(self):
        super().__init__()
        self.proj = nn.ModuleList([nn.Conv2d(3, c, 1) for c in (64, 64, 128, 256, 512)])

    def forward(self, x):
        outs = []
        for i, p in enumerate(self.proj):
            outs.append(p(F.avg_pool2d(x, 2 ** (i + 1))))
        return outs


def _model(H, W, k, D, dev):
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint

    m = DepthModelCVHint(H, W, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=k + 1)
    gu.set_formula_weights(m, 11)
    m.encoder = TinyImageEncoder()
    gu.set_formula_weights(m.encoder, 12)
    return m.to(dev).eval()


def _batch(f, b, k, H, W, dev, cams):
    Ks0, Kfull, cTw = cams
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    img = lambda s, n: tt(syn.hash_normalish((n, 3, H, W), s))
    sl = slice(f * b, f * b + b)
    cur = {"image_b3hw": img(100 + f, b), "K_s0_b44": tt(Ks0[sl]), "invK_s0_b44": tt(np.linalg.inv(Ks0[sl])),
           "K_s1_b44": tt(Ks0[sl] * np.array([0.5, 0.5, 1, 1], np.float32)[None, :, None]),
           "cam_T_world_b44": tt(cTw[sl]), "world_T_cam_b44": tt(np.linalg.inv(cTw[sl]).astype(np.float32)),
           "K_full_depth_b44": tt(Kfull[sl])}
    cur["invK_s1_b44"] = torch.linalg.inv(cur["K_s1_b44"])
    # source views: small translations of the current camera
    src_T = []
    for j in range(k):
        pose = np.linalg.inv(cTw[sl].astype(np.float64))
        pose[:, :3, 3] += np.array([0.05 * (j + 1), -0.03 * (j + 1), 0.0])
        src_T.append(np.linalg.inv(pose).astype(np.float32))
    src_T = np.stack(src_T, 1)
    src = {"image_b3hw": torch.stack([img(200 + 10 * f + j, b) for j in range(k)], 1),
           "K_s1_b44": cur["K_s1_b44"].unsqueeze(1).expand(b, k, 4, 4).contiguous(),
           "cam_T_world_b44": tt(src_T), "world_T_cam_b44": tt(np.linalg.inv(src_T).astype(np.float32))}
    return cur, src


def _cams(n, H2, W2):
    _, K, T = syn.tsdf_frames(1, H2, W2, seed=3, bounds=BD)
    pose0 = np.linalg.inv(T[0].astype(np.float64))
    cTw = []
    for f in range(n):
        pose = pose0.copy()
        pose[:3, 3] += np.array([0.02 * f, 0.01 * f, 0.0])
        cTw.append(np.linalg.inv(pose).astype(np.float32))
    Ks0 = np.repeat(K[:1], n, 0)
    Kfull = Ks0.copy()
    Kfull[:, :2] *= 2.0
    return Ks0, Kfull, np.stack(cTw)


def test_reference_shaped_forward_equals_forward_from_features():
    import gpu_util as gu
    from doubletake_amd.utils.rendering_utils import empty_hint

    dev = gu.dev()
    H, W, k, D, b = 128, 160, 3, 16, 2
    model = _model(H, W, k, D, dev)
    cur, src = _batch(0, b, k, H, W, dev, _cams(b, H // 2, W // 2))
    empty_hint(cur, torch.zeros(b, 1, H // 2, W // 2, device=dev))
    out = model("test", cur, src, return_mask=True)
    torch.cuda.synchronize()
    for i in range(4):
        d = out[f"depth_pred_s{i}_b1hw"]
        assert tuple(d.shape) == (b, 1, (H // 2) >> i, (W // 2) >> i) and torch.isfinite(d).all()
        assert torch.allclose(d, torch.exp(out[f"log_depth_pred_s{i}_b1hw"]), rtol=1e-6)
    assert tuple(out["lowest_cost_bhw"].shape) == (b, H // 4, W // 4) and tuple(out["overall_mask_bhw"].shape) == (b, k, H // 4, W // 4)
    # the same through the parts
    feats = model.encoder(cur["image_b3hw"])
    m_cur, m_src = model.compute_matching_feats(cur["image_b3hw"], src["image_b3hw"])
    ext = src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1)
    poses = cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"]
    want = model.forward_from_features(feats, m_cur, m_src, ext, poses, src["K_s1_b44"], cur["invK_s1_b44"], cur, return_mask=True)
    for key in want:
        assert torch.equal(out[key], want[key]), key


def test_offline_two_pass_on_one_gpu():
    import gpu_util as gu
    from doubletake_amd import loops, parallel
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    H, W, k, D, b, nb = 128, 160, 2, 16, 2, 3
    H2, W2 = H // 2, W // 2
    model = _model(H, W, k, D, dev)
    cams = _cams(b * nb, H2, W2)
    surface, _, _ = syn.tsdf_frames(1, H2, W2, seed=3, bounds=BD)
    base = torch.from_numpy(surface[0:1] * np.float32(0.55)).to(dev)
    seen = {"hint_cover": [], "first_depths": []}

    def model_fn(cur_data, src_data):
        out = model("test", cur_data, src_data, return_mask=True)
        # random-weight networks do not predict metric depth: fuse a plausible surface modulated by the prediction
        out["depth_pred_s0_b1hw"] = base + 0.02 * torch.tanh(out["depth_pred_s0_b1hw"] - 1.0)
        seen["hint_cover"].append(float(cur_data["depth_hint_mask_b1hw"].mean()))
        return out

    load = lambda i: _batch(i, b, k, H, W, dev, cams)
    first, between, second = loops.two_pass_fns(model_fn, load, (H2, W2), fuse_size=(H, W))
    hint_fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    final_fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    sf_hint = parallel.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=hint_fuser)
    sf_final = parallel.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=final_fuser)
    n1, n2 = parallel.run_two_pass(nb, lambda i: b, first, second, sf_hint, sf_final, between_passes=between)
    torch.cuda.synchronize()
    assert (n1, n2) == (b * nb, b * nb)
    # pass 1 ran with empty hints, pass 2 with rendered ones (the same three views fused several times -> weights above the cut)
    assert seen["hint_cover"][:nb] == [0.0] * nb and min(seen["hint_cover"][nb:]) > 0.2
    # the hint TSDF equals fusing pass 1 serially (world size 1: no exchange in between)
    check = OurFuser(None, 0.04, 3.0, bounds=BD)
    for i in range(nb):
        d, K, T = first(i)
        check.fuse_frames(d, K, T, None)
    a, c = hint_fuser.tsdf_fuser_pred.tsdf, check.tsdf_fuser_pred.tsdf
    assert torch.equal(a.tsdf_values.view(torch.int16), c.tsdf_values.view(torch.int16))
    assert (final_fuser.tsdf_fuser_pred.tsdf.tsdf_weights > 0).sum().item() > 5000
