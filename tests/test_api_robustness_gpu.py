"""Boundary behaviour of the drop-in modules: dtype / layout normalisation and loud failures."""
import numpy as np
import pytest
import torch

from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _setup(k=2, h=12, w=20, D=8, b=2):
    import gpu_util as gu
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    inp = syn.volume_inputs(b, k, h, w, 16, 3)
    t = gu.to_dev(inp)
    m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
    gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 31)
    gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 32)
    return gu, m, t


def test_volume_accepts_other_dtypes_and_strides():
    gu, m, t = _setup()
    args = gu.volume_call_args(t)
    hd = gu.hint_dict(t)
    ref, low, _, mask = m(**args, cv_depth_hint_dict=hd, return_mask=True)
    # non-contiguous source features (a permuted view of an NHWC buffer) and float64 camera matrices
    alt = dict(args)
    alt["src_feats"] = args["src_feats"].permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    assert not alt["src_feats"].is_contiguous()
    for name in ("src_extrinsics", "src_poses", "src_Ks", "cur_invK"):
        alt[name] = args[name].double()
    got, low2, _, mask2 = m(**alt, cv_depth_hint_dict=hd, return_mask=True)
    assert torch.equal(got, ref) and torch.equal(low2, low) and torch.equal(mask2, mask)
    # half features are promoted to fp32 (values exactly representable in half -> same result)
    h16 = dict(args)
    h16["cur_feats"] = args["cur_feats"].half()
    h16["src_feats"] = args["src_feats"].half()
    ref16, *_ = m(**{**args, "cur_feats": h16["cur_feats"].float(), "src_feats": h16["src_feats"].float()},
                  cv_depth_hint_dict=hd)
    got16, *_ = m(**h16, cv_depth_hint_dict=hd)
    assert torch.equal(got16, ref16)


def test_loud_failures():
    from doubletake_amd import _abi
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager

    gu, m, t = _setup()
    args = gu.volume_call_args(t)
    hd = gu.hint_dict(t)
    with pytest.raises(_abi.DoubletakeHipError):  # CPU tensors: no fallback
        m(**{k_: (v.cpu() if torch.is_tensor(v) else v) for k_, v in args.items()}, cv_depth_hint_dict=hd)
    with pytest.raises(ValueError):
        m(**{**args, "cur_feats": args["cur_feats"][:, :, :-1]}, cv_depth_hint_dict=hd)
    # more source views than the fused kernel keeps resident in LDS (7): up to 15 the MFMA kernel streams the further views'
    # layer-1 weights from L2 (round 3), beyond that the general HIP kernel takes over with a warning
    from oracle import cost_volume_ref as cref

    for k in (9, 12):
        inp = syn.volume_inputs(1, k, 8, 12, 16, 5)
        t9 = gu.to_dev(inp)
        m9 = FeatureMeshHintVolumeManager(8, 12, num_depth_bins=6, num_source_views=k).to(gu.dev())
        mw = gu.load_formula_mlp(m9.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 31)
        hw9 = gu.load_formula_mlp(m9.hint_mlp, [3, 12, 12, 1], 32)
        import warnings as _w

        with _w.catch_warnings():
            _w.simplefilter("error")  # no fallback warning: the fused kernel runs
            vol9 = m9(**gu.volume_call_args(t9), cv_depth_hint_dict=gu.hint_dict(t9))[0]
        simple9 = m9._forward_impl(**gu.volume_call_args(t9), cv_depth_hint_dict=gu.hint_dict(t9), depth_planes_bdhw=None,
                                   return_mask=False, _impl="simple")[0]
        want9, _, _ = cref.feature_volume(inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"],
                                          inp["cur_invK"], inp["min_depth"], inp["max_depth"], 6, mw,
                                          hint={n: inp[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")},
                                          hint_mlp_weights=hw9)
        assert np.abs(vol9.cpu().numpy() - want9).max() < 5e-5, k
        assert np.abs(simple9.cpu().numpy() - want9).max() < 5e-5, k
        # the no-hint manager through the same streamed instantiation
        from doubletake_amd.modules.cost_volume import FeatureVolumeManager

        f9 = FeatureVolumeManager(8, 12, num_depth_bins=6, num_source_views=k).to(gu.dev())
        gu.load_formula_mlp(f9.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 31)
        want_f, _, _ = cref.feature_volume(inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"],
                                           inp["cur_invK"], inp["min_depth"], inp["max_depth"], 6, mw)
        assert np.abs(f9(**gu.volume_call_args(t9))[0].cpu().numpy() - want_f).max() < 5e-5, k
    k = 16
    inp = syn.volume_inputs(1, k, 8, 12, 16, 5)
    t17 = gu.to_dev(inp)
    m17 = FeatureMeshHintVolumeManager(8, 12, num_depth_bins=6, num_source_views=k).to(gu.dev())
    mw = gu.load_formula_mlp(m17.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 31)
    hw17 = gu.load_formula_mlp(m17.hint_mlp, [3, 12, 12, 1], 32)
    with pytest.warns(UserWarning, match="source views"):
        vol17 = m17(**gu.volume_call_args(t17), cv_depth_hint_dict=gu.hint_dict(t17))[0]
    want17, _, _ = cref.feature_volume(inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"],
                                       inp["cur_invK"], inp["min_depth"], inp["max_depth"], 6, mw,
                                       hint={n: inp[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")},
                                       hint_mlp_weights=hw17)
    assert np.abs(vol17.cpu().numpy() - want17).max() < 5e-5
    # conv primitive: channel counts the MFMA tiling cannot express run on the general-shape kernel (same result as torch)
    conv = torch.nn.Conv2d(12, 20, 3, padding=1).to(gu.dev())
    xin = torch.from_numpy(syn.hash_normalish((1, 12, 8, 8), 3)).to(gu.dev())
    got = ops.conv2d([(ops.as_nhwc(xin), False)], conv)
    want = torch.nn.functional.conv2d(xin.cpu(), conv.weight.detach().cpu(), conv.bias.detach().cpu(), padding=1)
    assert (got.cpu() - want).abs().max().item() < 1e-5
    # ... but asking for the MFMA entry point directly with such a shape fails loudly, with a retrievable message
    d = _abi.ConvDesc()
    d.n, d.h_out, d.w_out, d.c_out, d.nsrc, d.ksize, d.stride, d.act, d.h_in, d.w_in = 1, 8, 8, 32, 1, 3, 1, 0, 8, 8
    d.c[0] = 12
    import ctypes

    buf = torch.zeros(4096, device=gu.dev())
    assert _abi.lib().dt_conv2d_f32(ctypes.byref(d), _abi.ptr(buf), None, None, _abi.ptr(buf), None, None, _abi.ptr(buf), None) != 0
    assert "multiple of 8" in _abi.lib().dt_last_error().decode()
    with pytest.raises(_abi.DoubletakeHipError):
        ops.as_nhwc(torch.zeros(1, 8, 4, 4))


def test_decoder_and_encoder_take_nchw_inputs():
    """Feature maps in the default NCHW layout are converted on entry; results equal the channels_last call."""
    import gpu_util as gu
    from doubletake_amd.modules.networks import CVEncoder

    enc = CVEncoder(num_ch_cv=8, num_ch_enc=[64, 128], num_ch_outs=[64, 128]).to(gu.dev())
    gu.set_formula_weights(enc, 3)
    vol = torch.from_numpy(syn.hash_normalish((1, 8, 16, 24), 1)).to(gu.dev())
    feats = [torch.from_numpy(syn.hash_normalish((1, 64, 16, 24), 2)).to(gu.dev()),
             torch.from_numpy(syn.hash_normalish((1, 128, 8, 12), 3)).to(gu.dev())]
    a = enc(vol, feats)
    b = enc(vol.contiguous(memory_format=torch.channels_last), [f.contiguous(memory_format=torch.channels_last) for f in feats])
    for x, y in zip(a, b):
        assert torch.equal(x, y) and x.shape[1] in (64, 128)
