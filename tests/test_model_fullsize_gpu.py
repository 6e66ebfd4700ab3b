"""GPU: the whole hot path (DepthModelCVHint.forward_from_features = mesh-hint volume -> CVEncoder -> decoder -> exp) at
every BASELINE.json shape -- cfg2 640x480, cfg3 512x384 batch 8, cfg4 512x384 batch 1, cfg5 portrait 384x512 with 96
planes; small (SkipDecoderRegression) and full (DepthDecoderPP) models -- against checksums and 256 probes per output
captured from the REFERENCE's own modules at full size (tests/golden/make_golden.py:gen_model_fullsize).
Tolerance on depth: the north star's 1e-3 abs."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

CASES = {
    "cfg2_small": (1, 7, 120, 160, 64, 201, "skip"),
    "cfg2_full": (1, 7, 120, 160, 64, 202, "unet_pp"),
    "cfg3_full_b8": (8, 7, 96, 128, 64, 203, "unet_pp"),
    "cfg3_small_b8": (8, 7, 96, 128, 64, 204, "skip"),
    "cfg4_small": (1, 7, 96, 128, 64, 205, "skip"),
    "cfg5_full_d96": (2, 7, 128, 96, 96, 206, "unet_pp"),
    "cfg5_small_d96": (2, 7, 128, 96, 96, 207, "skip"),
}
ENC = {"skip": ("resnet18d", [64, 64, 128, 256, 512]), "unet_pp": ("efficientnet", [24, 48, 64, 160, 256])}


def build_case(name):
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint

    b, k, h, w, D, seed, dec = CASES[name]
    enc_name, widths = ENC[dec]
    model = DepthModelCVHint(4 * h, 4 * w, image_encoder_name=enc_name, depth_decoder_name=dec, matching_num_depth_bins=D,
                             model_num_views=k + 1, matching_encoder_type=None)
    gu.set_formula_weights(model.cost_volume.mlp, seed + 1)
    gu.set_formula_weights(model.cost_volume.hint_mlp, seed + 2)
    gu.set_formula_weights(model.cost_volume_net, seed + 3)
    gu.set_formula_weights(model.depth_decoder, seed + 4)
    if dec == "unet_pp":
        # head gains / biases of the fixture (make_golden.py:gen_model_fullsize): log depth ~ zero mean, std 0.4 per scale,
        # so depth spans about 0.2 .. 6 m and the tolerances below are small against the signal
        g = load_golden("model_fullsize_checksums.npz")
        with torch.no_grad():
            for i in range(4):
                head = model.depth_decoder.convs[f"output_{i}"][1]
                head.weight.mul_(float(g[f"{name}|pp_head_gain"][i]))
                head.bias.fill_(float(g[f"{name}|pp_head_bias"][i]))
    model = model.to(gu.dev())
    inp = syn.volume_inputs(b, k, h, w, 16, seed)
    t = gu.to_dev(inp)
    pyr = [torch.from_numpy(p).to(gu.dev()) for p in syn.prior_pyramid(b, widths, 2 * h, 2 * w, seed + 50)]
    return model, inp, t, pyr


def run_case(name):
    import gpu_util as gu

    model, inp, t, pyr = build_case(name)
    out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                      t["cur_invK"], gu.hint_dict(t), return_mask=True)
    torch.cuda.synchronize()
    return out


def _check(g, name, key, got, atol, frac_ok=0.0):
    got = np.asarray(got, dtype=np.float32).reshape(-1)
    idx = g[f"{name}|{key}|probe_idx"]
    want = g[f"{name}|{key}|probe_val"]
    assert idx.max() < got.size, (name, key, got.size)
    d = np.abs(got[idx] - want)
    bad = (d > atol).mean()
    assert bad <= frac_ok, f"{name} {key}: {bad:.3%} of 256 probes differ by more than {atol} (max {d.max():.3e})"
    if frac_ok == 0.0:
        n = got.size
        assert abs(float(got.astype(np.float64).sum()) - float(g[f"{name}|{key}|sum"])) <= atol * n
        assert abs(float(np.abs(got.astype(np.float64)).sum()) - float(g[f"{name}|{key}|abssum"])) <= atol * n
        assert abs(float(got.min()) - float(g[f"{name}|{key}|min"])) <= 4 * atol
        assert abs(float(got.max()) - float(g[f"{name}|{key}|max"])) <= 4 * atol


@pytest.mark.parametrize("name", list(CASES))
def test_whole_model_against_reference_fullsize_checksums(name):
    g = load_golden("model_fullsize_checksums.npz")
    b, k, h, w, D, seed, dec = CASES[name]
    assert [int(v) for v in g[f"{name}|meta"]] == [b, k, h, w, D, seed]
    out = run_case(name)
    for i in range(4):
        ld = out[f"log_depth_pred_s{i}_b1hw"]
        assert tuple(ld.shape) == (b, 1, (2 * h) >> i, (2 * w) >> i)
        _check(g, name, f"log_depth_pred_s{i}_b1hw", ld.cpu().numpy(), 2e-4)
        _check(g, name, f"depth_pred_s{i}_b1hw", out[f"depth_pred_s{i}_b1hw"].cpu().numpy(), 1e-3)  # north-star tolerance
    # the fixture has real dynamic range (round 2's UNet++ cases spanned < 0.1 in log depth)
    assert float(g[f"{name}|depth_pred_s0_b1hw|max"]) / float(g[f"{name}|depth_pred_s0_b1hw|min"]) > 10.0
    # the argmax plane can flip between two nearly equal scores: allow isolated probes
    _check(g, name, "lowest_cost", out["lowest_cost_bhw"].cpu().numpy(), 1e-5, frac_ok=0.02)
    m = out["overall_mask_bhw"]
    assert tuple(m.shape) == (b, k, h, w)   # slow manager semantics: per-view masks of the last plane
    _check(g, name, "mask_sum", m.float().sum(1).cpu().numpy(), 0.5, frac_ok=0.01)


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_full_b8", "cfg5_full_d96"])
def test_volume_and_encoder_features_against_reference_fullsize_checksums(name):
    """The intermediate tensors of the same cases: cost volume and the four CVEncoder maps."""
    import gpu_util as gu

    g = load_golden("model_fullsize_checksums.npz")
    model, inp, t, pyr = build_case(name)
    md, Md = (torch.tensor(v, device=gu.dev()).view(1, 1, 1, 1) for v in (model.min_matching_depth, model.max_matching_depth))
    vol, low, _, _ = model.cost_volume(cur_feats=t["cur_feats"], src_feats=t["src_feats"], src_extrinsics=t["src_extrinsics"],
                                       src_poses=t["src_poses"], src_Ks=t["src_Ks"], cur_invK=t["cur_invK"], min_depth=md,
                                       max_depth=Md, return_mask=False, cv_depth_hint_dict=gu.hint_dict(t))
    _check(g, name, "volume", vol.contiguous().cpu().numpy(), 5e-5)
    feats = model.cost_volume_net(vol, pyr[1:])
    for i, f in enumerate(feats):
        _check(g, name, f"cv_feat{i}", f.contiguous().cpu().numpy(), 2e-4)


@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_full_b8"])
def test_whole_model_split_precision_mode_within_depth_tolerance(name):
    """Opt-in split-precision volume kernel (manager.precision = "split16") inside the whole model, against the same
    reference full-size checksums and the same 1e-3 depth tolerance as the exact-fp32 path."""
    import gpu_util as gu

    g = load_golden("model_fullsize_checksums.npz")
    model, inp, t, pyr = build_case(name)
    model.cost_volume.precision = "split16"
    out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                      t["cur_invK"], gu.hint_dict(t), return_mask=True)
    torch.cuda.synchronize()
    for i in range(4):
        _check(g, name, f"depth_pred_s{i}_b1hw", out[f"depth_pred_s{i}_b1hw"].cpu().numpy(), 1e-3)
        _check(g, name, f"log_depth_pred_s{i}_b1hw", out[f"log_depth_pred_s{i}_b1hw"].cpu().numpy(), 5e-4)


PP_NODES = ("in_conv_31", "in_conv_22", "in_conv_13", "in_conv_01", "in_conv_02", "in_conv_03", "in_conv_04")


def _run_decoder_nodes(name):
    """DepthDecoderPP alone on the CVEncoder maps of the case: (node outputs, head outputs, model, inputs)."""
    import gpu_util as gu

    model, inp, t, pyr = build_case(name)
    md, Md = (torch.tensor(v, device=gu.dev()).view(1, 1, 1, 1) for v in (model.min_matching_depth, model.max_matching_depth))
    vol, _, _, _ = model.cost_volume(cur_feats=t["cur_feats"], src_feats=t["src_feats"], src_extrinsics=t["src_extrinsics"],
                                     src_poses=t["src_poses"], src_Ks=t["src_Ks"], cur_invK=t["cur_invK"], min_depth=md,
                                     max_depth=Md, return_mask=False, cv_depth_hint_dict=gu.hint_dict(t))
    feats = [pyr[0]] + model.cost_volume_net(vol, pyr[1:])
    return model, feats


@pytest.mark.parametrize("name", ["cfg2_full", "cfg3_full_b8", "cfg5_full_d96"])
def test_unetpp_node_outputs_against_reference_fullsize_checksums(name):
    """The UNet++ decoder's node outputs X_ij (reference modules/networks.py:65-85, captured with forward hooks on
    convs['in_conv_ij'] at full size) -- what cv_feat{i} is for the encoder: a wrong skip / diagonal / up input of any
    node shows up here at O(0.5) against a 3e-4 tolerance, long before the heads."""
    g = load_golden("model_fullsize_checksums.npz")
    model, feats = _run_decoder_nodes(name)
    nodes = {}
    model.depth_decoder(feats, _nodes=nodes)
    torch.cuda.synchronize()
    assert set(PP_NODES) <= set(nodes)
    for n in PP_NODES:
        _check(g, name, f"pp_{n}", nodes[n].contiguous().cpu().numpy(), 3e-4)


def test_fullsize_fixture_catches_a_swapped_skip_connection():
    """Negative control: feed right_conv_02 -- the skip path of node X_03 -- the stale map
    X_01 instead of X_02 (a one-character wiring slip in DepthDecoderPP.forward).  The full-size checks must fail."""
    g = load_golden("model_fullsize_checksums.npz")
    name = "cfg2_full"
    model, feats = _run_decoder_nodes(name)
    dec = model.depth_decoder
    seen = {}
    r01, r02 = dec.convs["right_conv_01"], dec.convs["right_conv_02"]
    run01, run02 = r01.run, r02.run

    def rec(srcs, impl="mfma"):
        seen["x01"] = srcs
        return run01(srcs, impl=impl)

    r01.run = rec
    r02.run = lambda srcs, impl="mfma": run02(seen["x01"], impl=impl)
    try:
        nodes = {}
        out = dec(feats, _nodes=nodes)
        torch.cuda.synchronize()
    finally:
        del r01.run, r02.run
    _check(g, name, "pp_in_conv_02", nodes["in_conv_02"].contiguous().cpu().numpy(), 3e-4)  # upstream of the slip: still right
    with pytest.raises(AssertionError):
        _check(g, name, "pp_in_conv_03", nodes["in_conv_03"].contiguous().cpu().numpy(), 3e-4)
    with pytest.raises(AssertionError):
        _check(g, name, "log_depth_pred_s0_b1hw", out["log_depth_pred_s0_b1hw"].cpu().numpy(), 2e-4)
    with pytest.raises(AssertionError):
        _check(g, name, "depth_pred_s0_b1hw", torch.exp(out["log_depth_pred_s0_b1hw"]).cpu().numpy(), 1e-3)


@pytest.mark.parametrize("name", ["cfg2_small", "cfg2_full", "cfg3_full_b8"])
def test_whole_model_split_precision_conv_stack_within_depth_tolerance(name):
    """Opt-in split-precision conv stack (conv_ops.CONV_PRECISION = "split16": the 3x3 stride-1 layers on
    csrc/conv_wino_split.hip, fp16 hi/lo products with fp32 accumulation) inside the whole model, against the same reference
    full-size checksums and the same tolerances as the exact-fp32 path (depth 1e-3, log depth 2e-4, UNet++ nodes 3e-4)."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    g = load_golden("model_fullsize_checksums.npz")
    model, inp, t, pyr = build_case(name)
    prev = ops.CONV_PRECISION
    ops.CONV_PRECISION = "split16"
    try:
        before = int(ops._abi.lib().dt_kernel_launch_count())
        out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                          t["cur_invK"], gu.hint_dict(t), return_mask=True)
        torch.cuda.synchronize()
        assert int(ops._abi.lib().dt_kernel_launch_count()) > before
    finally:
        ops.CONV_PRECISION = prev
    for i in range(4):
        _check(g, name, f"depth_pred_s{i}_b1hw", out[f"depth_pred_s{i}_b1hw"].cpu().numpy(), 1e-3)
        _check(g, name, f"log_depth_pred_s{i}_b1hw", out[f"log_depth_pred_s{i}_b1hw"].cpu().numpy(), 2e-4)


def test_cfg2_small_whole_tensors_vs_torch_cpu_oracle():
    """VERDICT r4 "weak" #1: the full-size fixtures above are probes and checksums.  Here every element of every output of
    the headline configuration (640x480, 7 source views, 64 planes, DoubleTake-small) is compared with the torch-CPU
    restatement of the path (oracle/torch_cpu_ref.py -- pinned to the reference goldens, and measured bit-identical to the
    imported reference on this very kind of frame: tests/golden/cpu_ref_vs_port.py): cost volume 5e-5, lowest-cost planes
    exact up to near-ties, CVEncoder maps 2e-4, log depth 2e-4, depth within the north star's 1e-3.  About 8 s of CPU work."""
    import gpu_util as gu
    from oracle import torch_cpu_ref as tref

    model, inp, t, pyr = build_case("cfg2_small")
    D = CASES["cfg2_small"][4]
    cv_feats_gpu = []
    hook = model.cost_volume_net.register_forward_hook(lambda m, i, o: cv_feats_gpu.extend(o))
    vol_gpu = []
    hook2 = model.cost_volume.register_forward_hook(lambda m, i, o: vol_gpu.append(o[0]))
    try:
        out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                          t["cur_invK"], gu.hint_dict(t), return_mask=True)
        torch.cuda.synchronize()
    finally:
        hook.remove()
        hook2.remove()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    lin = lambda pre: [(sd[f"{pre}.net.{i}.weight"], sd[f"{pre}.net.{i}.bias"]) for i in (0, 2, 4)]
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    ti = {n: torch.from_numpy(v) for n, v in inp.items()}
    hint = {n: ti[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}
    prev = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        vol, planes = tref.hint_volume_loop(ti["cur_feats"], ti["src_feats"], ti["src_extrinsics"], ti["src_poses"], ti["src_Ks"],
                                            ti["cur_invK"], ti["min_depth"], ti["max_depth"], D, lin("cost_volume.mlp"), hint=hint,
                                            hint_mlp=lin("cost_volume.hint_mlp"))
        pyr_c = [p.cpu() for p in pyr]
        cv = tref.cv_encoder(vol, pyr_c[1:], sub("cost_volume_net."))
        dec = tref.skip_decoder_regression([pyr_c[0]] + cv, sub("depth_decoder."))
    finally:
        torch.set_num_threads(prev)
    v_gpu = vol_gpu[0].cpu()
    assert tuple(v_gpu.shape) == tuple(vol.shape) and float((v_gpu - vol).abs().max()) < 5e-5
    # lowest cost: the plane of the per-pixel maximum; a pixel may pick another plane only where two costs tie to 1e-4
    low = out["lowest_cost_bhw"].cpu()
    best = vol.max(1).values
    idx = (planes.view(planes.shape[0], -1, 1, 1) - low.unsqueeze(1)).abs().argmin(1, keepdim=True)
    assert float((best - vol.gather(1, idx).squeeze(1)).abs().max()) < 1e-4
    assert len(cv_feats_gpu) == len(cv)
    for a, b in zip(cv_feats_gpu, cv):
        assert float((a.cpu() - b).abs().max()) < 2e-4
    for i in range(4):
        ld = dec[f"log_depth_pred_s{i}_b1hw"]
        assert float((out[f"log_depth_pred_s{i}_b1hw"].cpu() - ld).abs().max()) < 2e-4
        assert float((out[f"depth_pred_s{i}_b1hw"].cpu() - torch.exp(ld)).abs().max()) < 1e-3


def _whole_tensor_case(name, element=None):
    """GPU outputs + intermediates of one full-size case and the torch-CPU oracle's on the same inputs (``element``: compare that
    batch element only -- the oracle then runs on it alone; the GPU runs the whole batch)."""
    import gpu_util as gu
    from oracle import torch_cpu_ref as tref

    model, inp, t, pyr = build_case(name)
    b, k, h, w, D, seed, dec = CASES[name]
    cv_gpu, vol_gpu, nodes_gpu = [], [], {}
    hook = model.cost_volume_net.register_forward_hook(lambda m, i, o: cv_gpu.extend(o))
    hook2 = model.cost_volume.register_forward_hook(lambda m, i, o: vol_gpu.append(o[0]))
    if dec == "unet_pp":
        orig = model.depth_decoder.forward
        model.depth_decoder.forward = lambda feats, **kw: orig(feats, _nodes=nodes_gpu, **kw)
    try:
        out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                          t["cur_invK"], gu.hint_dict(t), return_mask=True)
        torch.cuda.synchronize()
    finally:
        hook.remove()
        hook2.remove()
    sl = slice(None) if element is None else slice(element, element + 1)
    sd = {kk: v.detach().cpu() for kk, v in model.state_dict().items()}
    lin = lambda pre: [(sd[f"{pre}.net.{i}.weight"], sd[f"{pre}.net.{i}.bias"]) for i in (0, 2, 4)]
    sub = lambda pre: {kk[len(pre):]: v for kk, v in sd.items() if kk.startswith(pre)}
    ti = {n: torch.from_numpy(v)[sl] if (v.ndim and v.shape[0] == b) else torch.from_numpy(v) for n, v in inp.items()}
    hint = {n: ti[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}
    prev = torch.get_num_threads()
    torch.set_num_threads(max(8, prev))
    nodes_cpu = {}
    try:
        vol, planes = tref.hint_volume_loop(ti["cur_feats"], ti["src_feats"], ti["src_extrinsics"], ti["src_poses"], ti["src_Ks"],
                                            ti["cur_invK"], ti["min_depth"], ti["max_depth"], D, lin("cost_volume.mlp"), hint=hint,
                                            hint_mlp=lin("cost_volume.hint_mlp"))
        pyr_c = [p.cpu()[sl] for p in pyr]
        cv = tref.cv_encoder(vol, pyr_c[1:], sub("cost_volume_net."))
        if dec == "unet_pp":
            dd = tref.depth_decoder_pp([pyr_c[0]] + cv, sub("depth_decoder."), nodes=nodes_cpu)
        else:
            dd = tref.skip_decoder_regression([pyr_c[0]] + cv, sub("depth_decoder."))
    finally:
        torch.set_num_threads(prev)
    return dict(out=out, vol_gpu=vol_gpu[0], cv_gpu=cv_gpu, nodes_gpu=nodes_gpu, vol=vol, planes=planes, cv=cv, dec=dd,
                nodes=nodes_cpu, sl=sl)


def _assert_whole_tensors(c):
    sl = c["sl"]
    d = lambda a, b: float((a.cpu()[sl] - b).abs().max())
    assert tuple(c["vol_gpu"][sl].shape) == tuple(c["vol"].shape) and d(c["vol_gpu"], c["vol"]) < 5e-5
    assert len(c["cv_gpu"]) == len(c["cv"]) == 4
    for a, b in zip(c["cv_gpu"], c["cv"]):
        assert d(a, b) < 2e-4
    for i in range(4):
        ld = c["dec"][f"log_depth_pred_s{i}_b1hw"]
        assert d(c["out"][f"log_depth_pred_s{i}_b1hw"], ld) < 2e-4
        assert d(c["out"][f"depth_pred_s{i}_b1hw"], torch.exp(ld)) < 1e-3


def test_cfg2_full_whole_tensors_vs_torch_cpu_oracle():
    """DepthDecoderPP (F1: 292 GFLOP, the largest FLOP consumer of the full model) was pinned at full size
    by 256 probes and sums only.  Here EVERY element of the cost volume, the four CVEncoder maps, all ten UNet++ node outputs
    X_ij, the four log-depth maps and the four depth maps of the 640x480 full model is compared with the torch-CPU
    restatement (oracle/torch_cpu_ref.py, pinned to the reference goldens).  Tolerances: volume 5e-5, feature maps and
    UNet++ nodes 2e-4 (the nodes are O(1) activations), log depth 2e-4, depth 1e-3 (north star)."""
    c = _whole_tensor_case("cfg2_full")
    _assert_whole_tensors(c)
    assert sorted(c["nodes_gpu"]) == sorted(c["nodes"]) and len(c["nodes"]) == 10
    for name, want in c["nodes"].items():
        got = c["nodes_gpu"][name].cpu()
        assert tuple(got.shape) == tuple(want.shape), name
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 2e-4 * scale, (name, float((got - want).abs().max()), scale)


def test_cfg5_small_d96_whole_tensors_vs_torch_cpu_oracle():
    """the D = 96 portrait shape (384x512 image, 96x128 matching resolution, batch 2), one batch element,
    every element of every tensor: volume, CVEncoder maps, log depth, depth."""
    c = _whole_tensor_case("cfg5_small_d96", element=1)
    _assert_whole_tensors(c)


def test_cfg5_full_d96_whole_tensors_vs_torch_cpu_oracle():
    """BASELINE configs[4]'s shape with the full model (portrait 384x512, 96 planes, batch 2, DepthDecoderPP): one batch element,
    every element of the volume, the CVEncoder maps, all ten UNet++ nodes, log depth and depth against the torch-CPU restatement."""
    c = _whole_tensor_case("cfg5_full_d96", element=0)
    _assert_whole_tensors(c)
    assert len(c["nodes"]) == 10 and sorted(c["nodes_gpu"]) == sorted(c["nodes"])
    for name, want in c["nodes"].items():
        got = c["nodes_gpu"][name].cpu()[c["sl"]]
        assert tuple(got.shape) == tuple(want.shape), name
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) < 2e-4 * scale, (name, float((got - want).abs().max()), scale)
