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
    ext, poses = model.relative_poses(cur, src)   # one HIP launch instead of the reference's two torch.matmul calls
    assert torch.allclose(ext, src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1), atol=1e-6)
    assert torch.allclose(poses, cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"], atol=1e-6)
    want = model.forward_from_features(feats, m_cur, m_src, ext, poses, src["K_s1_b44"], cur["invK_s1_b44"], cur, return_mask=True)
    for key in want:
        assert torch.equal(out[key], want[key]), key


def test_model_with_other_matching_feature_dims_runs_on_the_general_volume_kernel():
    """opts.matching_feature_dims != 16 (sr_depth_model.py:200,207): matching encoder with 8 output channels + the general
    volume kernel; the volume inside the model equals the module called alone (pinned on the reference in test_volume_gpu)."""
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.utils.rendering_utils import empty_hint

    dev = gu.dev()
    H, W, k, D, b = 128, 160, 2, 8, 1
    m = DepthModelCVHint(H, W, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=k + 1, matching_feature_dims=8)
    gu.set_formula_weights(m, 11)
    m.encoder = TinyImageEncoder()
    gu.set_formula_weights(m.encoder, 12)
    m = m.to(dev).eval()
    assert m.cost_volume.matching_dim_size == 8 and m.cost_volume.mlp.net[0].in_features == syn.mlp_in_channels(k, 8)
    cur, src = _batch(0, b, k, H, W, dev, _cams(b, H // 2, W // 2))
    empty_hint(cur, torch.zeros(b, 1, H // 2, W // 2, device=dev))
    out = m("test", cur, src, return_mask=True)
    m_cur, m_src = m.compute_matching_feats(cur["image_b3hw"], src["image_b3hw"])
    assert tuple(m_cur.shape) == (b, 8, H // 4, W // 4) and tuple(m_src.shape) == (b, k, 8, H // 4, W // 4)
    torch.cuda.synchronize()
    assert all(torch.isfinite(out[f"depth_pred_s{i}_b1hw"]).all() for i in range(4))
    assert float(out["depth_pred_s0_b1hw"].std()) > 0


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


@pytest.mark.parametrize("volume_type,golden_key,tol", [("simple_cost_volume", "dot_volume", 5e-4), ("mlp_feature_volume", "mlp_volume", 5e-5)])
@pytest.mark.parametrize("case", ["k2_land", "k7_land"])
def test_depth_model_simple_and_mlp_volume_types(case, volume_type, golden_key, tol):
    """DepthModel (SimpleRecon, sr_depth_model.py:186-204) reaches the dot-product and the no-hint MLP managers through the
    model class: its volume equals the reference golden of that manager, and forward_from_features equals the chain of its
    parts and the numpy oracle of the conv stacks on that golden volume (depth within 1e-3)."""
    import gpu_util as gu
    from conftest import load_golden
    from doubletake_amd.experiment_modules.doubletake_model import DepthModel
    from oracle import networks_ref as nref

    dev = gu.dev()
    g = load_golden(f"volume_{case}.npz")
    b, k, h, w, D, seed, empty, behind = [int(v) for v in g["meta"]]
    inp = syn.volume_inputs(b, k, h, w, 16, seed, empty_hint=bool(empty), behind_view=bool(behind))
    t = gu.to_dev(inp)
    model = DepthModel(4 * h, 4 * w, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=k + 1,
                       min_matching_depth=float(inp["min_depth"].reshape(-1)[0]), max_matching_depth=float(inp["max_depth"].reshape(-1)[0]),
                       feature_volume_type=volume_type, matching_encoder_type=None)
    gu.set_formula_weights(model, 21)
    if volume_type == "mlp_feature_volume":
        gu.load_formula_mlp(model.cost_volume.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    model = model.to(dev).eval()
    pyr = syn.prior_pyramid(b, [64, 64, 128, 256, 512], 2 * h, 2 * w, 4)
    pyr_t = [torch.from_numpy(p).to(dev) for p in pyr]
    out = model.forward_from_features(pyr_t, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                      t["cur_invK"], return_mask=True)
    vol, low, _, mask = model.cost_volume(**gu.volume_call_args(t), return_mask=True)
    torch.cuda.synchronize()
    assert np.abs(vol.cpu().numpy() - g[golden_key]).max() < tol
    assert torch.equal(out["lowest_cost_bhw"], low)
    if volume_type == "simple_cost_volume":
        assert out["overall_mask_bhw"] is None and mask is None  # cost_volume.py: the dot-product manager returns no mask
    else:
        np.testing.assert_array_equal(out["overall_mask_bhw"].cpu().numpy(), g["mlp_mask_slow"])
    # conv stacks on the REFERENCE volume through the numpy oracle
    sd = {n: v.detach().cpu().numpy() for n, v in model.state_dict().items()}
    sub = lambda pre: {n[len(pre):]: v for n, v in sd.items() if n.startswith(pre)}
    cv = nref.cv_encoder(g[golden_key], pyr[1:], sub("cost_volume_net."))
    ref = nref.skip_decoder_regression([pyr[0]] + cv, sub("depth_decoder."))
    for i in range(4):
        got = out[f"depth_pred_s{i}_b1hw"].cpu().numpy()
        want = np.exp(ref[f"log_depth_pred_s{i}_b1hw"])
        assert np.abs(got - want).max() < 1e-3, (i, np.abs(got - want).max())


def _second_pass_hints_torch(state, cur_data, H2, W2, first_T_second=None):
    """The reference's op sequence (test_offline_two_pass.py:311-358 / test_revisit.py:194-231) with torch ops -- the form
    loops.py used in round 2; kept here as the checker of the fused hints_from_mesh."""
    dev = cur_data["cam_T_world_b44"].device
    b = cur_data["cam_T_world_b44"].shape[0]
    K = cur_data["K_s0_b44"].clone()
    K[:, 0] /= W2
    K[:, 1] /= H2
    pose = cur_data["world_T_cam_b44"].float()
    if first_T_second is not None:
        pose = first_T_second.to(dev).float() @ pose
    depth, _ = state["renderer"].render(state["mesh"], torch.inverse(pose).clone(), K)
    hint = depth.clone()
    hint[hint == -1] = float("nan")
    mask_b = ~torch.isnan(hint)
    ys, xs = torch.meshgrid(torch.arange(H2, device=dev), torch.arange(W2, device=dev), indexing="ij")
    pix = torch.stack([xs.flatten() + 0.5, ys.flatten() + 0.5, torch.ones(H2 * W2, device=dev)], 0)
    weights = []
    for j in range(b):
        cam = (cur_data["invK_s0_b44"][j, :3, :3].float() @ pix) * depth[j].reshape(1, -1)
        world = (pose[j] @ torch.cat([cam, torch.ones_like(cam[:1])], 0))[:3].t().contiguous()
        weights.append(state["hint_fuser"].sample_tsdf(world, what_to_sample="weights"))
    weights = torch.stack(weights, 0).view(b, 1, H2, W2).clone()
    weights[~mask_b] = 0.0
    return hint, mask_b, weights


def test_revisit_flow_and_fused_second_pass_hints_on_one_gpu(tmp_path):
    """loops.revisit_fns (reference test_revisit.py:104-260) through parallel.run_two_pass with world size 1: the previous
    scan is the same room in a shifted world frame, so (i) the hint TSDF holds the previous scan's frames, (ii) the hints
    the new scan sees through first_scan_T_second_scan equal the reference's op sequence evaluated with torch ops (fused
    kernel, cut disabled: weights within 2e-6, identical masks), with and without the rigid transform, (iii) the per-frame
    hint_time / model_time land in the reference's score sheet format."""
    import gpu_util as gu
    from doubletake_amd import loops, parallel
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.utils import formats

    dev = gu.dev()
    H, W, k, D, b, nb, nb1 = 128, 160, 2, 16, 2, 3, 2
    H2, W2 = H // 2, W // 2
    model = _model(H, W, k, D, dev)
    cams = _cams(b * nb, H2, W2)
    surface, _, _ = syn.tsdf_frames(1, H2, W2, seed=3, bounds=BD)
    base = torch.from_numpy(surface[0:1] * np.float32(0.55)).to(dev)
    shift = torch.tensor([[1, 0, 0, 0.08], [0, 1, 0, -0.04], [0, 0, 1, 0.02], [0, 0, 0, 1.0]], device=dev)[None]  # first_T_second
    seen = {"cover": []}

    def model_fn(cur_data, src_data):
        out = model("test", cur_data, src_data, return_mask=True)
        out["depth_pred_s0_b1hw"] = base + 0.02 * torch.tanh(out["depth_pred_s0_b1hw"] - 1.0)
        seen["cover"].append(float(cur_data["depth_hint_mask_b1hw"].mean()))
        return out

    load = lambda i: _batch(i, b, k, H, W, dev, cams)

    def load_first(i):  # previous scan: the same cameras expressed in ITS world frame (world_first = shift @ world_new)
        cur, src = load(i)
        cur["world_T_cam_b44"] = (shift @ cur["world_T_cam_b44"]).contiguous()
        cur["cam_T_world_b44"] = torch.inverse(cur["world_T_cam_b44"]).contiguous()
        src["world_T_cam_b44"] = (shift[:, None] @ src["world_T_cam_b44"]).contiguous()
        src["cam_T_world_b44"] = torch.inverse(src["world_T_cam_b44"]).contiguous()
        return cur, src

    checks = []

    def on_frame(i, cur_data, out):
        want = _second_pass_hints_torch(state_box["s"], load(i)[0], H2, W2, shift)
        checks.append((cur_data["depth_hint_b1hw"].clone(), cur_data["depth_hint_mask_b_b1hw"].clone(),
                       cur_data["sampled_weights_b1hw"].clone(), want))

    timer = loops.FrameTimer()
    first, between, second = loops.revisit_fns(model_fn, load_first, load, shift, (H2, W2), fuse_size=(H, W), on_frame=on_frame,
                                               timer=timer)
    state_box = {}

    def between_rec(f):
        state_box["s"] = between(f)
        return state_box["s"]

    bd_first = dict(xmin=-2.2, xmax=2.2, ymin=-2.2, ymax=2.2, zmin=-0.2, zmax=2.6)
    hint_fuser = OurFuser(None, 0.04, 3.0, bounds=bd_first)
    final_fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    sf_hint = parallel.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=hint_fuser)
    sf_final = parallel.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=final_fuser)
    n1, n2 = parallel.run_two_pass(nb, lambda i: b, first, second, sf_hint, sf_final, between_passes=between_rec,
                                   num_first_batches=nb1)
    torch.cuda.synchronize()
    assert (n1, n2) == (b * nb1, b * nb)
    assert seen["cover"][:nb1] == [0.0] * nb1 and min(seen["cover"][nb1:]) > 0.2
    assert len(checks) == nb
    for hint, mask_b, weights, (w_hint, w_mask, w_weights) in checks:
        assert torch.equal(mask_b, w_mask) and mask_b.dtype == torch.bool
        assert torch.equal(torch.isnan(hint), torch.isnan(w_hint))
        assert torch.equal(hint[mask_b], w_hint[w_mask])
        assert (weights - w_weights).abs().max().item() < 2e-6
        assert float(weights[~mask_b].abs().max()) == 0.0 and float(weights.max()) > 0.0  # zero outside the render
        # no 0.025 cut (test_offline_two_pass.py:354-356 has it commented out): rendered pixels keep their hint however
        # small the sampled weight is
        assert bool((mask_b & (weights < 0.025)).any()) or float(weights[mask_b].min()) >= 0.025
    # without the transform (= the two-pass second pass) the same equality holds
    cur, _ = load(0)
    st = between(hint_fuser)
    loops.hints_from_mesh(st["mesh"], hint_fuser, st["renderer"], cur, (H2, W2))
    w_hint, w_mask, w_weights = _second_pass_hints_torch(st, load(0)[0], H2, W2)
    assert torch.equal(cur["depth_hint_mask_b_b1hw"], w_mask) and (cur["sampled_weights_b1hw"] - w_weights).abs().max().item() < 2e-6
    # timing sheet (test_revisit.py:233-256 -> ResultsAverager JSON)
    pf = timer.per_frame()
    assert len(pf["hint_time"]) == len(pf["model_time"]) == b * nb and min(pf["model_time"]) > 0 and min(pf["hint_time"]) > 0
    doc = timer.write_scores(str(tmp_path / "scores.json"), "revisit", extra={"abs_diff": 0.1})
    back = formats.read_scores_json(str(tmp_path / "scores.json"))
    assert list(back["scores"]) == ["abs_diff", "hint_time", "model_time"] and back["scores"]["model_time"] == doc["scores"]["model_time"]


def test_hip_graph_replay_equals_eager_launches():
    """model.enable_hip_graphs(): forward_from_features and the single-image matching-encoder pass replayed from captured
    hipGraphs give bit-identical outputs to eager launches, frame after frame with changing inputs; a weight update
    invalidates the captured graphs; the feature-cache lookahead on a side stream feeds the same features."""
    import gpu_util as gu
    from doubletake_amd.utils.rendering_utils import empty_hint

    dev = gu.dev()
    H, W, k, D, b = 128, 160, 3, 16, 1
    model = _model(H, W, k, D, dev)
    cams = _cams(4, H // 2, W // 2)
    frames = []
    for f in range(4):
        cur, src = _batch(f, b, k, H, W, dev, cams)
        empty_hint(cur, torch.zeros(b, 1, H // 2, W // 2, device=dev))
        cur["depth_hint_b1hw"] = torch.full_like(cur["depth_hint_b1hw"], 1.0 + 0.1 * f)   # hints that differ per frame
        cur["depth_hint_mask_b1hw"] = torch.ones_like(cur["depth_hint_mask_b1hw"])
        cur["sampled_weights_b1hw"] = torch.full_like(cur["sampled_weights_b1hw"], 0.3)
        cur["frame_id_string"] = [f"{10 + f:06d}"]
        src["frame_id_string"] = [[f"{10 + f - 1 - i:06d}"] for i in range(k)]
        frames.append((cur, src))
    keys = [f"depth_pred_s{i}_b1hw" for i in range(4)] + ["lowest_cost_bhw", "overall_mask_bhw"]
    eager = [{kk: v.clone() for kk, v in model("test", c, s, return_mask=True).items() if kk in keys} for c, s in frames]
    model.enable_hip_graphs(True)
    for rep in range(2):
        for (c, s), want in zip(frames, eager):
            out = model("test", c, s, return_mask=True)
            for kk in keys:
                assert torch.equal(out[kk], want[kk]), (rep, kk)
    assert model._graphed_forward.captures == 1 and model._graphed_forward.replays == 8
    # weights change -> the graphs are rebuilt, results follow the new weights
    with torch.no_grad():
        model.depth_decoder.out4[4].bias.add_(0.25)
    out = model("test", *frames[0], return_mask=True)
    assert torch.allclose(out["log_depth_pred_s0_b1hw"], torch.log(eager[0]["depth_pred_s0_b1hw"]) + 0.25, atol=1e-5)
    assert model._graphed_forward.captures == 2 and model._graphed_forward.replays == 9   # (reset + one new capture)
    # feature cache + lookahead: the next frame's keyframe encoded on a side stream by the graphed encoder
    model.use_feature_cache = True
    model.matching_feature_cache.clear()
    ref_feats = model.matching_model(frames[1][0]["image_b3hw"])
    assert model.prefetch_matching_feats(frames[1][0]["image_b3hw"], frames[1][0]["frame_id_string"]) == 1
    assert model.prefetch_matching_feats(frames[1][0]["image_b3hw"], frames[1][0]["frame_id_string"]) == 0   # already cached
    m_cur, _ = model.compute_matching_feats(frames[1][0]["image_b3hw"], frames[1][1]["image_b3hw"], cur_ids=frames[1][0]["frame_id_string"],
                                            src_ids=frames[1][1]["frame_id_string"])
    torch.cuda.synchronize()
    assert torch.equal(m_cur, ref_feats)


def test_one_model_call_on_keyframes_of_different_scans_with_the_feature_cache():
    """What loops.IncrementalScanBatch asks of the model (round 6): ONE ``model("test", ...)`` call on a batch whose elements are
    the current keyframes of DIFFERENT scans -- per-element cameras, hints, frame ids and scan ids (a list), source features
    served from the HBM feature cache under (scan, frame) keys -- gives every element what the same frame gives in a batch-1
    call of its own scan (fp32 rounding of the batched conv plans: depth within 1e-4 relative), over two turns so that the second
    turn's sources are cache hits written by the first."""
    import gpu_util as gu
    from doubletake_amd import loops
    from doubletake_amd.utils.rendering_utils import empty_hint

    dev = gu.dev()
    H, W, k, D = 128, 160, 3, 16
    model = _model(H, W, k, D, dev)
    model.use_feature_cache = True
    cams = _cams(4, H // 2, W // 2)

    def frame(scan, f):
        cur, src = _batch(2 * scan + f, 1, k, H, W, dev, cams)   # (other images and another camera per scan and frame)
        empty_hint(cur, torch.zeros(1, 1, H // 2, W // 2, device=dev))
        cur["depth_hint_b1hw"] = torch.full_like(cur["depth_hint_b1hw"], 1.0 + 0.2 * scan + 0.1 * f)
        cur["depth_hint_mask_b1hw"] = torch.ones_like(cur["depth_hint_mask_b1hw"])
        cur["sampled_weights_b1hw"] = torch.full_like(cur["sampled_weights_b1hw"], 0.3 + 0.1 * scan)
        cur["frame_id_string"] = [f"{10 + f:06d}"]                 # the SAME frame ids in both scans: only the scan id differs
        cur["scan_id_string"] = f"scan{scan}"
        src["frame_id_string"] = [[f"{10 + f - 1 - i:06d}"] for i in range(k)]
        return cur, src

    keys = [f"depth_pred_s{i}_b1hw" for i in range(4)] + ["lowest_cost_bhw"]
    model.matching_feature_cache.clear()
    alone = {}
    for scan in range(2):
        for f in range(2):
            out = model("test", *frame(scan, f), return_mask=True)
            alone[scan, f] = {kk: out[kk].clone() for kk in keys}
    entries_alone = len(model.matching_feature_cache)
    model.matching_feature_cache.clear()
    for f in range(2):
        items = [frame(scan, f) for scan in range(2)]
        cur = loops._collate_frames([it[0] for it in items])
        src = loops._collate_frames([it[1] for it in items])
        assert cur["scan_id_string"] == ["scan0", "scan1"] and cur["image_b3hw"].shape[0] == 2 and src["image_b3hw"].shape[:2] == (2, k)
        out = model("test", cur, src, return_mask=True)
        for scan in range(2):
            for kk in keys:
                a, b = out[kk][scan:scan + 1], alone[scan, f][kk]
                rel = float(((a - b).abs() / b.abs().clamp_min(1.0)).max())
                assert a.shape == b.shape and rel <= 1e-4, (f, scan, kk, rel)
    assert len(model.matching_feature_cache) == entries_alone   # the same (scan, frame) entries, no cross-scan sharing
    # the two scans' predictions differ from each other (the test would also pass on a model that ignored its inputs otherwise)
    assert float((alone[0, 1]["depth_pred_s0_b1hw"] - alone[1, 1]["depth_pred_s0_b1hw"]).abs().max()) > 1e-3
