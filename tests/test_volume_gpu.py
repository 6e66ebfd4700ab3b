"""GPU parity of the plane-sweep cost volume kernels (through the C ABI) against
 (1) golden vectors captured from the reference, (2) the numpy oracle on seeded inputs,
 (3) full-size (cfg1 / cfg2) checksums captured from the reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

CASES = ["k2_land", "k7_land", "k7_b2", "k3_portrait", "k7_empty", "k2_ragged"]


def _case(case):
    import gpu_util as gu

    g = load_golden(f"volume_{case}.npz")
    b, k, h, w, D, seed, empty, behind = [int(v) for v in g["meta"]]
    inp = syn.volume_inputs(b, k, h, w, 16, seed, empty_hint=bool(empty), behind_view=bool(behind))
    return g, inp, gu.to_dev(inp), (b, k, h, w, D, seed)


def _check_lowest(vol, low, planes_bd, atol):
    """lowest_cost must be a plane whose cost is within atol of the per-pixel maximum."""
    b, D = planes_bd.shape
    vmax = vol.max(1)
    idx = np.abs(planes_bd.reshape(b, D, 1, 1) - low[:, None]).argmin(1)
    picked = np.take_along_axis(vol, idx[:, None], 1)[:, 0]
    assert np.all(picked >= vmax - atol)


@pytest.mark.parametrize("case", CASES)
def test_dot_volume_vs_reference_golden(case):
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager

    g, inp, t, (b, k, h, w, D, seed) = _case(case)
    m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
    vol, low, planes, mask = m(**gu.volume_call_args(t))
    torch.cuda.synchronize()
    assert mask is None and tuple(vol.shape) == (b, D, h, w) and tuple(planes.shape) == (b, D, h, w)
    np.testing.assert_allclose(planes[:, :, 0, 0].cpu().numpy(), g["planes"], rtol=3e-6)
    np.testing.assert_allclose(vol.cpu().numpy(), g["dot_volume"], atol=5e-4, rtol=0)
    _check_lowest(g["dot_volume"], low.cpu().numpy(), g["planes"], 1e-3)


@pytest.mark.parametrize("impl", ["simple", "mfma"])
@pytest.mark.parametrize("case", CASES)
def test_hint_volume_vs_reference_golden(case, impl):
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    g, inp, t, (b, k, h, w, D, seed) = _case(case)
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 77 + seed)
    vol, low, planes, mask = m._forward_impl(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t),
                                             depth_planes_bdhw=None, return_mask=True, _impl=impl)
    torch.cuda.synchronize()
    v = vol.cpu().numpy()
    err = np.abs(v - g["hint_volume"]).max()
    assert err < 5e-5, f"{impl} max abs err {err}"
    np.testing.assert_array_equal(mask.cpu().numpy(), g["hint_mask_slow"])
    _check_lowest(g["hint_volume"], low.cpu().numpy(), g["planes"], 1e-4)
    # fast manager: same volume, any_k mask
    f = m.to_fast()
    vol2, _, _, mask2 = f(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t), return_mask=True)
    torch.cuda.synchronize()
    assert np.abs(vol2.cpu().numpy() - g["hint_volume_fast"]).max() < 5e-5
    np.testing.assert_array_equal(mask2.cpu().numpy(), g["hint_mask_fast"])


@pytest.mark.parametrize("case", CASES)
def test_mlp_volume_no_hint_vs_reference_golden(case):
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureVolumeManager

    g, inp, t, (b, k, h, w, D, seed) = _case(case)
    m = FeatureVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    vol, low, planes, mask = m(**gu.volume_call_args(t), return_mask=True)
    torch.cuda.synchronize()
    assert np.abs(vol.cpu().numpy() - g["mlp_volume"]).max() < 5e-5
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mlp_mask_slow"])
    np.testing.assert_array_equal(vol.contiguous().cpu().numpy(), vol.cpu().numpy())  # channels_last view is coherent


def _probe_check(vol, prefix, g, atol):
    flat = vol.reshape(-1)
    idx = g[f"{prefix}_probe_idx"]
    np.testing.assert_allclose(flat[idx], g[f"{prefix}_probe_val"], atol=atol, rtol=0)
    assert abs(flat.astype(np.float64).sum() - g[f"{prefix}_sum"]) < atol * flat.size * 0.05 + 1e-2
    assert abs(np.abs(flat.astype(np.float64)).sum() - g[f"{prefix}_abssum"]) < atol * flat.size * 0.05 + 1e-2


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2"])
def test_fullsize_checksums_from_reference(cfg):
    """BASELINE.json configs[0] (320x256, K2, D32) and configs[1] (640x480, K7, D64) volume shapes."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager, FeatureMeshHintVolumeManager

    g = load_golden("volume_fullsize_checksums.npz")
    b, k, h, w, D, seed = [int(v) for v in g[f"{cfg}_meta"]]
    inp = syn.volume_inputs(b, k, h, w, 16, seed)
    t = gu.to_dev(inp)
    m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
    vol, _, _, _ = m(**gu.volume_call_args(t))
    _probe_check(vol.cpu().numpy(), f"{cfg}_dot", g, 1e-3)
    hm = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(hm.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    gu.load_formula_mlp(hm.hint_mlp, [3, 12, 12, 1], 77 + seed)
    vol, _, _, _ = hm(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t))
    _probe_check(vol.contiguous().cpu().numpy(), f"{cfg}_hint", g, 1e-4)


def test_cfg1_vs_oracle_full_tensor():
    """Whole-tensor comparison with the numpy oracle at BASELINE configs[0] size."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager
    from oracle import cost_volume_ref as ref

    b, k, h, w, D, seed = 1, 2, 64, 80, 32, 7
    inp = syn.volume_inputs(b, k, h, w, 16, seed)
    t = gu.to_dev(inp)
    hm = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    mw = gu.load_formula_mlp(hm.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 5)
    hw_ = gu.load_formula_mlp(hm.hint_mlp, [3, 12, 12, 1], 6)
    vol, low, _, _ = hm(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t))
    want, planes, _ = ref.feature_volume(
        inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"], inp["cur_invK"],
        inp["min_depth"], inp["max_depth"], D, mw, hint=gu.hint_dict(inp), hint_mlp_weights=hw_)
    got = vol.cpu().numpy()
    assert np.abs(got - want).max() < 5e-5
    _check_lowest(want, low.cpu().numpy(), planes, 1e-4)


def test_linearity_in_last_layer_at_full_size():
    """Size-independent property at cfg2: scaling W3/b3 by c scales the (no-hint) volume by c."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureVolumeManager

    b, k, h, w, D = 1, 7, 120, 160, 64
    t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 9))
    m = FeatureVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
    v1 = m(**gu.volume_call_args(t))[0].clone()
    with torch.no_grad():
        m.mlp.net[4].weight.mul_(2.0)
        m.mlp.net[4].bias.mul_(2.0)
    v2 = m(**gu.volume_call_args(t))[0]
    torch.cuda.synchronize()
    assert torch.isfinite(v1).all()
    assert (v2 - 2.0 * v1).abs().max().item() < 1e-5


@pytest.mark.parametrize("name,b,k,h,w,D", [
    ("cfg3_batch8", 8, 7, 96, 128, 64),          # BASELINE configs[2]: ScanNet default 512x384, batch 8
    ("cfg5_portrait_D96", 2, 7, 128, 96, 96),    # BASELINE configs[4]: 3RScan rotated (portrait), 96 planes
    ("tiny_one_view_D5", 3, 1, 3, 5, 5),         # fewer pixels than one MFMA tile, one source view, D % 8 != 0
    ("ragged_D13", 1, 4, 7, 45, 13),             # partial tiles + partial store chunks
    ("odd_50x70_b2", 2, 3, 50, 70, 8),           # >= 64 tiles whose 32 pixels wrap image rows: the column-strip tile order
    ("odd_33x97", 1, 2, 33, 97, 16),             # ... with a last tile that is partly outside the image
])
def test_other_baseline_configs_mfma_vs_simple_and_oracle_probes(name, b, k, h, w, D):
    """Larger BASELINE shapes: the fused MFMA kernel against the independent one-thread-per-pair GPU
    kernel on the whole tensor, and against the numpy oracle on one batch element / 6 planes."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager
    from oracle import cost_volume_ref as ref

    inp = syn.volume_inputs(b, k, h, w, 16, 21)
    t = gu.to_dev(inp)
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    mw = gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 31)
    hw_ = gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 32)
    args = dict(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t), depth_planes_bdhw=None, return_mask=True)
    va, la, pa, ma = m._forward_impl(**args, _impl="mfma")
    vb, lb, pb, mb = m._forward_impl(**args, _impl="simple")
    torch.cuda.synchronize()
    assert tuple(va.shape) == (b, D, h, w)
    assert (va - vb).abs().max().item() < 5e-5
    assert torch.equal(ma, mb) and tuple(ma.shape) == (b, k, h, w)
    # oracle on the last batch element, a few planes (full oracle run at this size takes minutes)
    planes = ref.generate_depth_planes(inp["min_depth"], inp["max_depth"], D)
    np.testing.assert_allclose(pa[:, :, 0, 0].cpu().numpy(), planes, rtol=3e-6)
    bi = b - 1
    sel = lambda a: a[bi:bi + 1]
    hd = nearest = None
    for d in (0, 1, D // 2, D - 2, D - 1):
        feats, z, mask, pix = ref.mlp_input_features(sel(inp["cur_feats"]), sel(inp["src_feats"]), sel(inp["src_extrinsics"]),
                                                     sel(inp["src_poses"]), sel(inp["src_Ks"]), sel(inp["cur_invK"]), planes[bi:bi + 1, d])
        s = ref.mlp_forward(feats.reshape(h * w, -1), mw).reshape(h * w)
        hdm = ref.nearest_resize(sel(inp["depth_hint_b1hw"]), h, w).reshape(-1)
        hwm = ref.nearest_resize(sel(inp["sampled_weights_b1hw"]), h, w).reshape(-1).copy()
        hmm = ref.nearest_resize(sel(inp["depth_hint_mask_b1hw"]), h, w).reshape(-1) != 0
        hwm[~hmm] = 0
        with np.errstate(invalid="ignore"):
            hmap = np.where(hmm, np.abs(hdm - planes[bi, d]), np.float32(-1)).astype(np.float32)
        want = ref.mlp_forward(np.stack([s, hmap, hwm], -1), hw_).reshape(h, w)
        got = va[bi, d].cpu().numpy()
        assert np.abs(got - want).max() < 5e-5, (name, d)


def test_generate_depth_planes_public_method():
    """cost_volume.py:96-130 as a stand-alone call: equals the oracle's planes and the planes forward() returns."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager
    from oracle import cost_volume_ref as ref

    inp = syn.volume_inputs(2, 2, 12, 20, 16, 3)
    t = gu.to_dev(inp)
    m = CostVolumeManager(12, 20, num_depth_bins=16).to(gu.dev())
    planes = m.generate_depth_planes(2, t["min_depth"], t["max_depth"])
    assert tuple(planes.shape) == (2, 16, 12, 20)
    want = ref.generate_depth_planes(inp["min_depth"], inp["max_depth"], 16)
    np.testing.assert_allclose(planes[:, :, 0, 0].cpu().numpy(), want, rtol=3e-6)
    _, _, fwd_planes, _ = m(**gu.volume_call_args(t))
    assert torch.equal(planes, fwd_planes)


def test_warp_features_public_method_vs_oracle():
    """cost_volume.py:132-217 as a stand-alone call (the fused kernels do the same warp internally)."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager
    from oracle import cost_volume_ref as ref

    b, k, h, w, c = 2, 3, 19, 27, 16
    inp = syn.volume_inputs(b, k, h, w, c, 7)
    t = gu.to_dev(inp)
    m = CostVolumeManager(h, w, num_depth_bins=8).to(gu.dev())
    plane = np.array([1.3, 2.1], dtype=np.float32)
    depth = torch.from_numpy(np.broadcast_to(plane.reshape(b, 1, 1, 1), (b, 1, h, w)).copy()).to(gu.dev())
    world, depths, warped, mask = m.warp_features(t["src_feats"], t["src_extrinsics"], t["src_Ks"], t["cur_invK"], depth,
                                                  b, k, c, None)
    rw, rz, rwarp, rmask, _ = ref.warp_features(inp["src_feats"], inp["src_extrinsics"], inp["src_Ks"], inp["cur_invK"], plane)
    assert tuple(world.shape) == (b * k, 4, h * w) and tuple(warped.shape) == (b, k, c, h, w)
    np.testing.assert_allclose(world.cpu().numpy(), rw, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(depths.cpu().numpy().reshape(b, k, -1), rz, rtol=2e-6, atol=1e-6)
    assert np.array_equal(mask.cpu().numpy().reshape(b, k, -1), rmask)
    assert np.abs(warped.cpu().numpy().reshape(b, k, c, -1) - rwarp).max() < 2e-4


# ---- LDS-staged dot-product volume (csrc/cv_dot_lds.hip) -------------------------------------------------------
def _dot_with(impl, m, t):
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager

    CostVolumeManager._dot_impl = impl
    try:
        out = m(**gu.volume_call_args(t))
        torch.cuda.synchronize()
    finally:
        CostVolumeManager._dot_impl = "lds"
    return out


@pytest.mark.parametrize("name,b,k,h,w,D,behind", [
    ("cfg2", 1, 7, 120, 160, 64, False),           # BASELINE configs[1]
    ("cfg3_batch8", 8, 7, 96, 128, 64, False),     # configs[2]
    ("cfg5_portrait_D96", 2, 7, 128, 96, 96, False),
    ("behind_view", 1, 7, 24, 32, 8, True),        # a source camera behind the planes: un-boxable views go direct
    ("ragged", 2, 3, 19, 27, 5, True),             # partial tiles, D not a multiple of the plane group
    ("tiny", 1, 1, 3, 5, 5, False),
])
def test_dot_volume_lds_staged_equals_direct_bitwise(name, b, k, h, w, D, behind):
    """The staged path must give the SAME BITS as sampling every tap from global memory (identical expressions in
    identical order), no tap may miss its staged box, and at the BASELINE sizes the bulk of the work is staged."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager

    inp = syn.volume_inputs(b, k, h, w, 16, 77, behind_view=behind)
    t = gu.to_dev(inp)
    m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
    v_lds, low_lds, _, _ = _dot_with("lds", m, t)
    v_dir, low_dir, _, _ = _dot_with("direct", m, t)
    assert torch.equal(v_lds.view(torch.int32), v_dir.view(torch.int32))
    assert torch.equal(low_lds, low_dir)
    v_st, _, _, _ = _dot_with("stats", m, t)
    assert torch.equal(v_st.view(torch.int32), v_dir.view(torch.int32))
    staged, direct, empty, stray = (int(x) for x in m.last_dot_stats.tolist())
    assert stray == 0
    assert staged + direct + empty > 0
    if not behind and h * w >= 96 * 96:
        # far planes (several planes per box) are staged; single-plane boxes of the near planes go direct by design
        assert staged >= 0.3 * (staged + direct), (staged, direct, empty)


def test_dot_volume_lds_vs_oracle_whole_tensor_cfg1():
    """cfg1 (320x256, K=2, D=32) whole tensor against the numpy oracle (seconds on the CPU)."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager
    from oracle import cost_volume_ref as ref

    b, k, h, w, D = 1, 2, 64, 80, 32
    inp = syn.volume_inputs(b, k, h, w, 16, 5)
    t = gu.to_dev(inp)
    m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
    vol, low, planes, _ = m(**gu.volume_call_args(t))
    want, _ = ref.dot_cost_volume(inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_Ks"], inp["cur_invK"],
                                  inp["min_depth"], inp["max_depth"], D)
    np.testing.assert_allclose(vol.cpu().numpy(), want, atol=5e-4, rtol=0)


# ---- opt-in split-precision MLP volume (csrc/cv_mlp_split.hip) ---------------------------------------------------------------
@pytest.mark.parametrize("case", CASES)
def test_split_precision_volume_vs_reference_golden(case):
    """fp16 hi/lo operands on the fp16 matrix pipe, fp32 accumulation: its own parity budget (2e-4, the fp32 kernel's is
    5e-5) against the SAME reference goldens, hint and no-hint managers."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager, FeatureVolumeManager

    g, inp, t, (b, k, h, w, D, seed) = _case(case)
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    m.precision = "split16"
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 77 + seed)
    vol, low, planes, mask = m(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t), return_mask=True)
    torch.cuda.synchronize()
    err = np.abs(vol.cpu().numpy() - g["hint_volume"]).max()
    assert err < 2e-4, f"split16 hint volume: max abs err {err}"
    np.testing.assert_array_equal(mask.cpu().numpy(), g["hint_mask_slow"])
    m2 = FeatureVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    m2.precision = "split16"
    gu.load_formula_mlp(m2.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    vol2 = m2(**gu.volume_call_args(t))[0]
    torch.cuda.synchronize()
    assert np.abs(vol2.cpu().numpy() - g["mlp_volume"]).max() < 2e-4


def test_split_precision_volume_fullsize_vs_fp32_kernel_and_reference_checksums():
    """cfg2 (640x480, K=7, D=64): split16 against the exact-fp32 kernel on the whole tensor and against the reference's
    full-size checksums."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    g = load_golden("volume_fullsize_checksums.npz")
    b, k, h, w, D, seed = [int(v) for v in g["cfg2_meta"]]
    inp = syn.volume_inputs(b, k, h, w, 16, seed)
    t = gu.to_dev(inp)
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 77 + seed)
    exact = m(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t))[0]
    m.precision = "split16"
    split = m(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t))[0]
    torch.cuda.synchronize()
    d = (exact - split).abs()
    assert d.max().item() < 2e-4 and d.mean().item() < 5e-6, (d.max().item(), d.mean().item())
    _probe_check(split.contiguous().cpu().numpy(), "cfg2_hint", g, 2e-4)


# ---- shapes outside the tuned kernels (general one-thread-per-sample kernels): reference-generated fixture -------------
@pytest.mark.parametrize("name", ["pp_k3", "pp_k2_b2", "c8_k2", "c24_k3_pp"])
def test_per_pixel_planes_and_other_channel_counts_vs_reference(name):
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import CostVolumeManager, FeatureMeshHintVolumeManager, FeatureVolumeManager
    from test_oracle_volume import variant_case

    g, inp, (b, k, h, w, D, C, seed), planes = variant_case(name)
    t = gu.to_dev(inp)
    args = gu.volume_call_args(t)
    if planes is not None:
        args["depth_planes_bdhw"] = torch.from_numpy(planes).to(gu.dev())

    def lowest_ok(low, key):
        if planes is not None:   # the gather of the given planes is exact wherever the argmax is unambiguous
            vol = g[key + "_volume"]
            top2 = np.sort(vol, 1)[:, -2:]
            clear = (top2[:, 1] - top2[:, 0]) > 1e-4
            np.testing.assert_array_equal(low.cpu().numpy()[clear], g[key + "_lowest"][clear])
            assert clear.mean() > 0.5   # (the behind-view case has large tied regions in the dot volume)
        else:
            np.testing.assert_allclose(low.cpu().numpy(), g[key + "_lowest"], rtol=3e-6)

    m = CostVolumeManager(h, w, num_depth_bins=D).to(gu.dev())
    vol, low, used, _ = m(**args)
    np.testing.assert_allclose(vol.cpu().numpy(), g["dot_volume"], atol=5e-4, rtol=0)
    if planes is not None:
        np.testing.assert_array_equal(used.cpu().numpy(), planes)
    if name != "c8_k2":   # (its dot volume has exact ties at the image border: zero rows)
        lowest_ok(low, "dot")

    fv = FeatureVolumeManager(h, w, num_depth_bins=D, matching_dim_size=C, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(fv.mlp, [syn.mlp_in_channels(k, C), 128, 128, 1], 11 + seed)
    vol, low, _, mask = fv(**args, return_mask=True)
    assert np.abs(vol.cpu().numpy() - g["mlp_volume"]).max() < 5e-5
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mlp_mask"])
    lowest_ok(low, "mlp")

    hv = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, matching_dim_size=C, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(hv.mlp, [syn.mlp_in_channels(k, C), 128, 128, 1], 11 + seed)
    gu.load_formula_mlp(hv.hint_mlp, [3, 12, 12, 1], 77 + seed)
    vol, low, _, mask = hv(**args, cv_depth_hint_dict=gu.hint_dict(t), return_mask=True)
    assert np.abs(vol.cpu().numpy() - g["hint_volume"]).max() < 5e-5
    np.testing.assert_array_equal(mask.cpu().numpy(), g["hint_mask_slow"])
    lowest_ok(low, "hint")
    fast = hv.to_fast()
    vol2, _, _, mask2 = fast(**args, cv_depth_hint_dict=gu.hint_dict(t), return_mask=True)
    assert np.abs(vol2.cpu().numpy() - vol.cpu().numpy()).max() == 0
    if "hint_mask_fast" in g:
        assert np.abs(vol2.cpu().numpy() - g["hint_volume_fast"]).max() < 5e-5
        np.testing.assert_array_equal(mask2.cpu().numpy(), g["hint_mask_fast"])
    torch.cuda.synchronize()


def test_uniform_plane_override_still_takes_the_tuned_kernel():
    """depth_planes_bdhw that is constant over the image (what generate_depth_planes returns) stays on the fused kernel."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    g, inp, t, (b, k, h, w, D, seed) = _case("k7_land")
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 11 + seed)
    gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 77 + seed)
    planes = torch.from_numpy(g["planes"]).to(gu.dev()).view(b, D, 1, 1).expand(b, D, h, w).contiguous()
    vol, low, used, _ = m(**gu.volume_call_args(t), cv_depth_hint_dict=gu.hint_dict(t), depth_planes_bdhw=planes)
    torch.cuda.synchronize()
    assert m._planes_px is None and vol.is_contiguous(memory_format=torch.channels_last)
    assert np.abs(vol.cpu().numpy() - g["hint_volume"]).max() < 5e-5


def test_cost_aware_span_plan_only_moves_boundaries():
    """Round 4: dt_cv_mlp_hint_planned_f32 prices every (tile, plane) unit by the source views it can see and gives every wave
    a span of equal estimated work.  The plan must cover the unit space exactly once (bounds start at 0, end at the unit count,
    never decrease), price units between the all-empty and the all-visible cost, and leave the volume what the un-planned
    call computes (to the ulp level of a differently contracted multiply-add) -- at cfg2 size, at batch 2 with a view behind the camera, and on a ragged image."""
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    for (b, k, h, w, D, seed) in ((1, 7, 120, 160, 64, 1), (2, 3, 48, 64, 32, 7), (1, 2, 19, 27, 16, 5)):
        t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, seed))
        m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
        gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
        gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 4)
        args, hd = gu.volume_call_args(t), gu.hint_dict(t)
        m.use_span_plan = False
        ref = m(**args, cv_depth_hint_dict=hd)[0].clone()
        m.use_span_plan = True
        got = m(**args, cv_depth_hint_dict=hd)[0]
        torch.cuda.synchronize()
        # (not bitwise: the first plane of a span goes through a separately inlined copy of the view set-up whose multiply-adds
        #  the compiler may contract differently -- 1-2 ulp; every unit must have been computed, by exactly this arithmetic)
        assert float((got - ref).abs().max()) < 2e-6, (b, k, h, w, D)
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        units = b * ((h * w + 31) // 32) * D
        blocks = min(cus, (units + 7) // 8)
        n_ints = (cus * 8 + 2) // 2 * 2
        plan = m._last_plan.cpu().numpy()
        bounds = plan[: 4 * n_ints].view(np.int32)[: blocks * 8 + 1]
        ngroups = (units + 255) // 256
        pref = plan[4 * (n_ints + (ngroups + 1) // 2 * 2):][: 4 * units].view(np.uint32).astype(np.int64)
        cost = pref - np.where(np.arange(units) % 256 == 0, 0, np.roll(pref, 1))
        assert bounds[0] == 0 and bounds[-1] == units and np.all(np.diff(bounds) >= 0), (bounds[:4], bounds[-4:], units)
        # price = 290 + 27 K + 24 per visible view (+ 200 on the first plane of a tile: the plane-invariant contraction)
        assert cost.min() >= 290 + 27 * k and cost.max() <= 290 + 27 * k + 24 * k + 200
        first_plane = np.arange(units) % D == 0
        assert cost[first_plane].min() >= 290 + 27 * k + 200 and cost[~first_plane].max() <= 290 + 27 * k + 24 * k
        if h * w >= 19200:  # the bench frame: spans really differ in length, their estimated work does not
            work = np.add.reduceat(cost.astype(np.int64), bounds[:-1].clip(max=units - 1))
            work[np.diff(bounds) == 0] = 0
            older, younger = work.reshape(blocks, 8)[:, :4], work.reshape(blocks, 8)[:, 4:]
            assert older.std() / older.mean() < 0.06 and younger.std() / younger.mean() < 0.08
            pair = older + younger  # waves w and w + 4 share a SIMD: the younger one's span compensates its partner's rounding
            assert pair.std() / pair.mean() < 0.012, pair.std() / pair.mean()
            assert np.diff(bounds).max() > np.diff(bounds).min() + 2


def test_span_plan_scratch_size_is_checked():
    """dt_cv_mlp_hint_planned_f32 refuses a plan scratch smaller than dt_cv_mlp_plan_bytes says (a buffer sized for another
    shape must fail the call, not be overrun): error return with a message, nothing launched, the next good call still works."""
    import gpu_util as gu
    from doubletake_amd import _abi
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    b, k, h, w, D = 1, 2, 24, 32, 16
    t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 11))
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
    gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 4)
    args, hd = gu.volume_call_args(t), gu.hint_dict(t)
    L = _abi.lib()
    need = int(L.dt_cv_mlp_plan_bytes(b, h, w, D))
    assert need > 0 and int(L.dt_cv_mlp_plan_bytes(0, h, w, D)) == 0
    good = m(**args, cv_depth_hint_dict=hd)[0].clone()
    orig = L.dt_cv_mlp_plan_bytes

    class Short:  # the module asks the library for the size: make it allocate one byte less
        def __getattr__(self, n):
            return (lambda *a: need - 1) if n == "dt_cv_mlp_plan_bytes" else getattr(L, n)

    real_lib = _abi.lib
    _abi.lib = lambda: Short()
    try:
        with pytest.raises(_abi.DoubletakeHipError, match="plan scratch"):
            m(**args, cv_depth_hint_dict=hd)
    finally:
        _abi.lib = real_lib
    assert L.dt_cv_mlp_plan_bytes is orig
    again = m(**args, cv_depth_hint_dict=hd)[0]
    torch.cuda.synchronize()
    assert torch.equal(again, good)


def test_cu_budget_only_changes_how_units_are_dealt():
    """Round 5: dt_cv_mlp_set_cu_budget(n) launches the persistent volume kernel with n workgroups (rounded down to a multiple
    of 8, at least 8, at most the device's CU count; 0 = the whole device).  Same volume whatever the budget -- planned and
    un-planned spans, hint and no-hint kernel -- and the setter reports the budget in force."""
    import gpu_util as gu
    from doubletake_amd import _abi
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager, FeatureVolumeManager

    L = _abi.lib()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    b, k, h, w, D = 1, 7, 60, 80, 32
    t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, 9))
    try:
        assert L.dt_cv_mlp_set_cu_budget(0) == cus
        for cls, kw in ((FeatureMeshHintVolumeManager, dict(cv_depth_hint_dict=gu.hint_dict(t))), (FeatureVolumeManager, {})):
            m = cls(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
            gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
            if kw:
                gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 4)
            assert L.dt_cv_mlp_set_cu_budget(0) == cus
            ref = m(**gu.volume_call_args(t), **kw)[0].clone()
            for budget, want in ((cus // 2, cus // 2 // 8 * 8), (13, 8), (100, 96), (10 * cus, cus)):
                assert L.dt_cv_mlp_set_cu_budget(budget) == want
                got = m(**gu.volume_call_args(t), **kw)[0]
                torch.cuda.synchronize()
                assert float((got - ref).abs().max()) < 2e-6, (cls.__name__, budget)
    finally:
        assert L.dt_cv_mlp_set_cu_budget(0) == cus
