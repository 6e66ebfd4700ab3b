"""GPU parity of the fused conv primitive and the encoder/decoder graphs built on it."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

H0, W0, D = 16, 24, 8


def _t(a):
    import gpu_util as gu

    return torch.from_numpy(np.ascontiguousarray(a)).to(gu.dev())


def _np(t):
    return t.contiguous().cpu().numpy()


CONV_CASES = [
    # (name, n, [(c, up)], h_in, w_in, cout, k, stride, act, residual)
    ("k3s1", 2, [(16, False)], 14, 18, 32, 3, 1, 1, False),
    ("k3s1_res_elu", 1, [(24, False)], 9, 13, 64, 3, 1, 2, True),
    ("k3s2", 2, [(16, False)], 14, 18, 32, 3, 2, 0, False),
    ("k3s2_odd", 1, [(40, False)], 15, 21, 96, 3, 2, 1, False),
    ("k1", 1, [(72, False)], 11, 7, 128, 1, 1, 2, False),
    ("cat2", 1, [(64, False), (48, False)], 12, 16, 64, 3, 1, 1, False),
    ("cat2_up", 2, [(32, True), (24, False)], 12, 16, 32, 3, 1, 2, False),
    ("cat3", 1, [(64, False), (64, False), (64, False)], 8, 12, 64, 3, 1, 1, True),
    ("k1_cat", 1, [(8, False), (8, True)], 6, 10, 32, 1, 1, 0, False),
    # deep-K / few-block layers take the 16-way K-split path; many-block layers the no-split path
    ("k3s1_split16", 1, [(256, False)], 8, 8, 32, 3, 1, 1, True),
    ("k3s2_split16", 1, [(192, False), (64, False)], 16, 16, 64, 3, 2, 2, False),
    ("k1_split16", 1, [(512, False)], 7, 9, 32, 1, 1, 0, False),
    ("k3s1_nosplit", 2, [(16, False)], 128, 160, 64, 3, 1, 1, True),
    ("k3s2_nosplit", 1, [(8, False)], 260, 252, 32, 3, 2, 0, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_primitive_mfma_vs_simple_vs_oracle(case):
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from oracle import networks_ref as ref

    name, n, srcs, h, w, cout, k, st, act, use_res = case
    cin = sum(c for c, _ in srcs)
    conv = nn.Conv2d(cin, cout, k, stride=st, padding=k // 2).to(gu.dev())
    gu.set_formula_weights(conv, 31 + cin)
    xs_np, xs = [], []
    for i, (c, up) in enumerate(srcs):
        hh, ww = (h // 2, w // 2) if up else (h, w)
        a = syn.hash_normalish((n, c, hh, ww), 100 + i)
        xs_np.append(a.repeat(2, 2).repeat(2, 3) if up else a)
        xs.append((ops.as_nhwc(_t(a)), up))
    ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    res_np = syn.hash_normalish((n, cout, ho, wo), 7) if use_res else None
    res = ops.as_nhwc(_t(res_np)) if use_res else None
    want = ref.conv2d(np.concatenate(xs_np, 1), conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(), stride=st)
    if use_res:
        want = want + res_np
    want = {0: lambda v: v, 1: ref.lrelu, 2: ref.elu}[act](want)
    got_s = _np(ops.conv2d(xs, conv, act=act, residual=res, impl="simple"))
    got_m = _np(ops.conv2d(xs, conv, act=act, residual=res, impl="mfma"))
    assert got_m.shape == want.shape
    assert np.abs(got_s - want).max() < 2e-5, f"simple {np.abs(got_s - want).max()}"
    assert np.abs(got_m - want).max() < 2e-5, f"mfma {np.abs(got_m - want).max()}"


def test_head_upsample_exp_layout():
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from oracle import networks_ref as ref

    x = syn.hash_normalish((2, 64, 9, 11), 3)
    xn = ops.as_nhwc(_t(x))
    np.testing.assert_array_equal(_np(xn), x)  # NCHW -> NHWC kernel is a pure permutation
    head = nn.Conv2d(64, 1, 1).to(gu.dev())
    gu.set_formula_weights(head, 8)
    want = ref.conv2d(x, head.weight.detach().cpu().numpy(), head.bias.detach().cpu().numpy())
    np.testing.assert_allclose(_np(ops.conv1x1_head(xn, head)), want, atol=2e-6)
    np.testing.assert_allclose(_np(ops.upsample2x_bilinear(xn)), ref.upsample_bilinear2(x), atol=1e-6)
    np.testing.assert_allclose(_np(ops.exp(_t(x))), np.exp(x), rtol=2e-6)


@pytest.mark.parametrize("name,cin,cout,stride", [("bb_same", 16, 16, 1), ("bb_chg", 24, 16, 1), ("bb_s2", 16, 32, 2)])
def test_basic_block_vs_reference_golden(name, cin, cout, stride):
    import gpu_util as gu
    from doubletake_amd.modules.layers import BasicBlock

    # (c_out = 16 does not fit the 32-channel MFMA tile: conv_ops routes those layers to the general-shape kernel)
    g = load_golden("networks.npz")
    blk = BasicBlock(cin, cout, stride=stride).to(gu.dev())
    gu.set_formula_weights(blk, 500 + cin + cout)
    x = syn.hash_normalish((2, cin, 14, 18), 900 + cin)
    np.testing.assert_allclose(_np(blk(_t(x))), g[f"{name}_out"], atol=3e-5, rtol=0)


@pytest.mark.parametrize("name,cin,cout,stride", [("bb_same", 16, 16, 1), ("bb_chg", 24, 16, 1), ("bb_s2", 16, 32, 2)])
def test_basic_block_simple_kernel_vs_reference_golden(name, cin, cout, stride):
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules.layers import BasicBlock

    g = load_golden("networks.npz")
    blk = BasicBlock(cin, cout, stride=stride).to(gu.dev())
    gu.set_formula_weights(blk, 500 + cin + cout)
    x = syn.hash_normalish((2, cin, 14, 18), 900 + cin)
    out = blk.run([(ops.as_nhwc(_t(x)), False)], impl="simple")
    np.testing.assert_allclose(_np(out), g[f"{name}_out"], atol=3e-5, rtol=0)


def test_small_model_graph_vs_reference_golden():
    """CVEncoder + SkipDecoderRegression (DoubleTake-small) against the reference's outputs."""
    import gpu_util as gu
    from doubletake_amd.modules.networks import CVEncoder
    from doubletake_amd.modules.networks_fast import SkipDecoderRegression

    g = load_golden("networks.npz")
    enc = [64, 64, 128, 256, 512]
    cve = CVEncoder(D, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 1234)
    vol = _t(syn.hash_normalish((1, D, H0, W0), 4321))
    feats = [_t(f) for f in syn.prior_pyramid(1, enc, 2 * H0, 2 * W0, 555)]
    outs = cve(vol, feats[1:])
    for i, o in enumerate(outs):
        np.testing.assert_allclose(_np(o), g[f"cve_small_out{i}"], atol=1e-4, rtol=0)
    dec = SkipDecoderRegression([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(dec, 2345)
    dout = dec([feats[0]] + outs)
    for k, v in dout.items():
        np.testing.assert_allclose(_np(v), g[f"skip_{k}"], atol=3e-4, rtol=0)


def test_full_model_graph_vs_reference_golden():
    """CVEncoder + DepthDecoderPP (full DoubleTake) against the reference's outputs."""
    import gpu_util as gu
    from doubletake_amd.modules.networks import CVEncoder, DepthDecoderPP

    g = load_golden("networks.npz")
    enc = [24, 48, 64, 160, 256]
    cve = CVEncoder(D, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 3456)
    vol = _t(syn.hash_normalish((1, D, H0, W0), 4321))
    feats = [_t(f) for f in syn.prior_pyramid(1, enc, 2 * H0, 2 * W0, 666)]
    outs = cve(vol, feats[1:])
    for i, o in enumerate(outs):
        np.testing.assert_allclose(_np(o), g[f"cve_full_out{i}"], atol=1e-4, rtol=0)
    dpp = DepthDecoderPP([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(dpp, 4567, scale_mult=0.7)
    dout = dpp([feats[0]] + outs)
    assert set(dout) == {f"log_depth_pred_s{i}_b1hw" for i in range(4)}
    for k, v in dout.items():
        np.testing.assert_allclose(_np(v), g[f"pp_{k}"], atol=5e-4, rtol=0)


def test_small_model_graph_full_size_mfma_vs_simple():
    """BASELINE configs[1] size (matching res 120x160, D=64): two independent GPU implementations
    (MFMA implicit GEMM vs one-thread-per-output direct conv) must agree on every output."""
    import gpu_util as gu
    from doubletake_amd.modules.networks import CVEncoder
    from doubletake_amd.modules.networks_fast import SkipDecoderRegression

    enc = [64, 64, 128, 256, 512]
    h, w, Dp = 120, 160, 64
    cve = CVEncoder(Dp, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 11)
    dec = SkipDecoderRegression([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(dec, 12)
    vol = _t(syn.hash_normalish((1, Dp, h, w), 1))
    feats = [_t(f) for f in syn.prior_pyramid(1, enc, 2 * h, 2 * w, 2)]
    a = dec([feats[0]] + cve(vol, feats[1:]))
    b = dec([feats[0]] + cve(vol, feats[1:], _impl="simple"), _impl="simple")
    torch.cuda.synchronize()
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape)
        err = (a[k] - b[k]).abs().max().item()
        assert err < 2e-4, f"{k}: {err}"
    assert tuple(a["log_depth_pred_s0_b1hw"].shape) == (1, 1, 2 * h, 2 * w)


@pytest.mark.parametrize("with_exp", [False, True])
def test_coarse_heads_one_launch_is_bit_identical(with_exp):
    """dt_head_mlp_multi_f32 (the three coarse regression heads of one decoder pass in ONE grid) against three
    dt_head_mlp_f32 launches: same body, so every output must be bit-equal; includes a ragged last tile."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    shapes = [(256, 30, 40), (128, 60, 80), (64, 120, 160), (64, 7, 9)]
    heads, xs = [], []
    for i, (c, h, w) in enumerate(shapes):
        head = torch.nn.Sequential(torch.nn.Conv2d(c, 128, 1), torch.nn.ELU(), torch.nn.Conv2d(128, 128, 1), torch.nn.ELU(),
                                   torch.nn.Conv2d(128, 1, 1)).to(gu.dev())
        gu.set_formula_weights(head, 40 + i)
        heads.append(head)
        xs.append(ops.as_nhwc(_t(syn.hash_normalish((1, c, h, w), 50 + i))))
    for n in (2, 3, 4):
        got = ops.head_mlp_multi(xs[:n], heads[:n], with_exp=with_exp)
        for x, head, g in zip(xs[:n], heads[:n], got):
            want = ops.head_mlp(x, head, with_exp=with_exp)
            if with_exp:
                assert torch.equal(g[0], want[0]) and torch.equal(g[1], want[1])
            else:
                assert torch.equal(g, want)
    # a map beyond the one-tile-per-workgroup limit is refused, not silently truncated
    big = ops.as_nhwc(torch.zeros((1, 64, 240, 320), device=gu.dev()))
    with pytest.raises(RuntimeError, match="dt_head_mlp_multi_f32"):
        ops.head_mlp_multi([xs[2], big], [heads[2], heads[3]])


def test_full_model_graph_full_size_mfma_vs_simple():
    """DoubleTake full model decoder (DepthDecoderPP, EfficientNetV2-S widths) at 120x160 / D=64."""
    import gpu_util as gu
    from doubletake_amd.modules.networks import CVEncoder, DepthDecoderPP

    enc = [24, 48, 64, 160, 256]
    h, w, Dp = 120, 160, 64
    cve = CVEncoder(Dp, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 21)
    dec = DepthDecoderPP([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(dec, 22, scale_mult=0.7)
    vol = _t(syn.hash_normalish((1, Dp, h, w), 3))
    feats = [_t(f) for f in syn.prior_pyramid(1, enc, 2 * h, 2 * w, 4)]
    a = dec([feats[0]] + cve(vol, feats[1:]))
    b = dec([feats[0]] + cve(vol, feats[1:], _impl="simple"), _impl="simple")
    torch.cuda.synchronize()
    for k in a:
        err = (a[k] - b[k]).abs().max().item()
        assert err < 5e-4, f"{k}: {err}"
    assert tuple(a["log_depth_pred_s0_b1hw"].shape) == (1, 1, 2 * h, 2 * w)


def test_batched_conv_graph_equals_per_sample():
    """Batch 3 through the small-model graph equals three batch-1 runs (n is just another tile index);
    not bit-equal because the K-split policy -- hence the fp32 summation order -- depends on the block count."""
    import gpu_util as gu
    from doubletake_amd.modules.networks import CVEncoder
    from doubletake_amd.modules.networks_fast import SkipDecoderRegression

    enc = [64, 64, 128, 256, 512]
    h, w, Dp = 24, 40, 16
    cve = CVEncoder(Dp, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 5)
    dec = SkipDecoderRegression([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(dec, 6)
    vol = _t(syn.hash_normalish((3, Dp, h, w), 8))
    feats = [_t(f) for f in syn.prior_pyramid(3, enc, 2 * h, 2 * w, 9)]
    full = dec([feats[0]] + cve(vol, feats[1:]))
    for i in range(3):
        one = dec([feats[0][i:i + 1]] + cve(vol[i:i + 1], [f[i:i + 1] for f in feats[1:]]))
        for k in full:
            assert (full[k][i:i + 1] - one[k]).abs().max().item() < 2e-5, k


@pytest.mark.parametrize("shape", [
    (1, 16, 32, 13, 21, "zeros", 1, False),     # ragged extent, one source
    (2, 24, 64, 16, 32, "zeros", 2, True),      # two sources, the second nearest-upsampled, residual + bias
    (1, 64, 32, 24, 40, "replicate", 1, False),  # replicate padding
    (1, 64, 64, 120, 160, "zeros", 1, True),    # 300 blocks on 256 CUs: tail split (256 whole blocks + 44 x 4 quarter workgroups)
    (1, 128, 64, 120, 160, "zeros", 2, True),   # the same with two sources (one nearest-upsampled)
])
def test_winograd_conv_vs_direct(shape):
    """conv_wino_kernel (F(2x2,3x3) on MFMA) against the direct one-thread-per-output conv."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    n, cin, cout, h, w, pad_mode, nsrc, with_res = shape
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, padding_mode=pad_mode).to(gu.dev())
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(syn.hash_normalish(tuple(conv.weight.shape), 5) * 0.1))
        conv.bias.copy_(torch.from_numpy(syn.hash_normalish((cout,), 6) * 0.1))
    if nsrc == 1:
        srcs = [(ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, cin, h, w), 1)).to(gu.dev())), False)]
    else:
        c0 = cin - 8
        srcs = [(ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, c0, h, w), 1)).to(gu.dev())), False),
                (ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, 8, h // 2, w // 2), 2)).to(gu.dev())), True)]
    res = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, cout, h, w), 3)).to(gu.dev())) if with_res else None
    want = ops.conv2d(srcs, conv, act=ops.ACT_LRELU02, residual=res, impl="simple")
    got = ops.conv2d(srcs, conv, act=ops.ACT_LRELU02, residual=res, impl="wino")
    assert got.shape == want.shape
    # fp32 Winograd vs fp32 direct: a few ulp of the largest output (2e-5 absolute for the O(1) outputs of the small cases)
    assert (got - want).abs().max().item() < 4e-6 * max(want.abs().max().item(), 5.0)
    again = ops.conv2d(srcs, conv, act=ops.ACT_LRELU02, residual=res, impl="wino")
    assert torch.equal(got, again)  # the cross-workgroup reduction sums in part order: run-to-run bit equality


@pytest.mark.parametrize("cin,cout,stride,h,w", [
    (128, 64, 1, 120, 160),   # Winograd (4 waves) + plain 1x1
    (256, 128, 1, 60, 80),    # Winograd K-split (8 waves) + K-split 1x1
    (512, 256, 1, 30, 40),    # direct 8-way K-split + K-split 1x1
    (896, 384, 1, 15, 20),    # same at the coarsest level (1x1 would pick the 16-way split on its own)
    (64, 128, 2, 120, 160),   # stride 2: 3x3 s2 + 3x3 s2 shortcut
    (256, 384, 2, 30, 40),
    (24, 32, 1, 10, 12),      # small shapes: combinations without a common workgroup size fall back to two launches
])
def test_paired_conv_launch_equals_two_launches(cin, cout, stride, h, w):
    """BasicBlock's conv1 + shortcut conv in one launch (dt_conv2d_pair_f32) against the two separate launches."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules.layers import BasicBlock

    blk = BasicBlock(cin, cout, stride=stride).to(gu.dev())
    gu.set_formula_weights(blk, 77 + cin)
    x = ops.as_nhwc(_t(syn.hash_normalish((1, cin, h, w), 5)))
    # two sources (concat) for the stride-1 cases, as CVEncoder's conv_i.0 uses it
    srcs = [(x[:, : cin // 2 // 8 * 8].contiguous(memory_format=torch.channels_last), False),
            (x[:, cin // 2 // 8 * 8:].contiguous(memory_format=torch.channels_last), False)] if stride == 1 else [(x, False)]
    a1, b1 = ops.conv2d_pair(srcs, blk.conv1, ops.ACT_LRELU02, blk.downsample[0], ops.ACT_NONE)
    a2 = ops.conv2d(srcs, blk.conv1, act=ops.ACT_LRELU02)
    b2 = ops.conv2d(srcs, blk.downsample[0], act=ops.ACT_NONE)
    torch.cuda.synchronize()
    if h * w > 120 * 160 or h * w < 15 * 20:
        assert torch.equal(a1, a2) and torch.equal(b1, b2)
    else:
        # the K split across waves / workgroups is planned per LAUNCH (a pair shares the CUs between its two
        # convolutions; a lone 300-block launch splits its leftover blocks), i.e. the two paths may sum the same fp32
        # products in a different order
        assert (a1 - a2).abs().max().item() < 4e-6 * max(a2.abs().max().item(), 5.0)
        assert (b1 - b2).abs().max().item() < 4e-6 * max(b2.abs().max().item(), 5.0)
    assert a1.abs().max().item() > 0.01 and b1.abs().max().item() > 0.01
    # same launch twice: bit-identical (the cross-workgroup reduction sums in part order, not in arrival order)
    a3, b3 = ops.conv2d_pair(srcs, blk.conv1, ops.ACT_LRELU02, blk.downsample[0], ops.ACT_NONE)
    assert torch.equal(a1, a3) and torch.equal(b1, b3)


@pytest.mark.parametrize("case", [
    # cin, cout, k, stride, h_in, w_in, pad_mode, up2 (second source nearest-upsampled), residual
    (384, 384, 3, 1, 15, 20, "zeros", False, True),       # transposed tiling (10 instead of 12 pixel tiles) + 2 workgroups per block
    (384, 256, 3, 1, 15, 20, "replicate", False, False),  # replicate padding in the transposed frame
    (256, 384, 3, 2, 30, 40, "zeros", False, False),      # stride 2 onto a 15x20 map
    (640, 256, 3, 1, 16, 12, "zeros", True, False),       # portrait (cfg5 coarse level), upsampled + plain source
    (896, 384, 1, 1, 15, 20, "zeros", False, False),      # 1x1 K-split kernel, transposed tiling
    (256, 256, 3, 1, 16, 32, "zeros", False, True),       # 128 blocks: cross-workgroup split without transposition
    (256, 256, 3, 1, 30, 40, "zeros", False, True),       # 320 blocks: stays one workgroup per block
])
def test_low_resolution_conv_tiling_and_cross_workgroup_split(case):
    """The 15x20 / 30x40 levels: tall-patch (transposed) tiling and the K split across workgroups (csrc/conv.hip) against
    the one-thread-per-output direct kernel, plus run-to-run bit equality under concurrent launches on two streams."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    cin, cout, k, st, h, w, pad_mode, up2, with_res = case
    conv = torch.nn.Conv2d(cin, cout, k, stride=st, padding=k // 2, padding_mode=pad_mode).to(gu.dev())
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(syn.hash_normalish(tuple(conv.weight.shape), 15) * (2.0 / (cin * k * k)) ** 0.5))
        conv.bias.copy_(torch.from_numpy(syn.hash_normalish((cout,), 16) * 0.1))
    if up2:
        c0 = cin - 384
        srcs = [(ops.as_nhwc(_t(syn.hash_normalish((1, 384, h // 2, w // 2), 1))), True),
                (ops.as_nhwc(_t(syn.hash_normalish((1, c0, h, w), 2))), False)]
    else:
        srcs = [(ops.as_nhwc(_t(syn.hash_normalish((1, cin, h, w), 1))), False)]
    ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
    res = ops.as_nhwc(_t(syn.hash_normalish((1, cout, ho, wo), 3))) if with_res else None
    want = ops.conv2d(srcs, conv, act=ops.ACT_LRELU02, residual=res, impl="simple")
    got = ops.conv2d(srcs, conv, act=ops.ACT_LRELU02, residual=res)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (1, cout, ho, wo)
    err = (got - want).abs().max().item()
    assert err < 3e-5, err
    assert got.abs().max().item() > 0.05
    # determinism under load: the same launch from two streams at once (separate scratch per stream), many times
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    outs = []
    for i in range(12):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            outs.append(ops.conv2d(srcs, conv, act=ops.ACT_LRELU02, residual=res))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, got)


def _random_conv_cases(n_cases=28, seed=20260927):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < n_cases:
        k = int(rng.choice([1, 3, 3, 3]))
        st = int(rng.choice([1, 1, 2])) if k == 3 else 1
        n = int(rng.choice([1, 1, 2, 3]))
        h, w = 2 * int(rng.randint(3, 21)), 2 * int(rng.randint(3, 21))  # even: an upsampled source needs it
        if rng.rand() < 0.3:
            h, w = h + 1, w + 1  # odd extents (then no upsampled source)
        cout = 32 * int(rng.randint(1, 13))
        nsrc = int(rng.choice([1, 1, 2, 3]))
        chans = [8 * int(rng.randint(1, 40)) for _ in range(nsrc)]
        ups = [bool(rng.rand() < 0.4) and h % 2 == 0 and w % 2 == 0 for _ in range(nsrc)]
        cases.append(dict(k=k, st=st, n=n, h=h, w=w, cout=cout, chans=chans, ups=ups, pad=str(rng.choice(["zeros", "replicate"])),
                          res=bool(rng.rand() < 0.5), act=int(rng.choice([0, 1, 2, 3])), seed=int(rng.randint(1, 1 << 20))))
    return cases


@pytest.mark.parametrize("c", _random_conv_cases(), ids=lambda c: f"k{c['k']}s{c['st']}_n{c['n']}_{c['h']}x{c['w']}_{'+'.join(map(str, c['chans']))}to{c['cout']}")
def test_conv_random_shapes_vs_direct(c):
    """Seeded sweep over shapes the planner routes differently (direct / Winograd, in-workgroup and cross-workgroup K split,
    transposed tiling, tail split, batch > 1, one to three sources with nearest-upsampled ones, both paddings, residual,
    every activation): the product path against the one-thread-per-output direct kernel, and run-to-run bit equality."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    cin = sum(c["chans"])
    conv = torch.nn.Conv2d(cin, c["cout"], c["k"], stride=c["st"], padding=c["k"] // 2, padding_mode=c["pad"]).to(gu.dev())
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(syn.hash_normalish(tuple(conv.weight.shape), c["seed"]) * (1.0 / (cin * c["k"] ** 2)) ** 0.5))
        conv.bias.copy_(torch.from_numpy(syn.hash_normalish((c["cout"],), c["seed"] + 1) * 0.1))
    srcs = []
    for i, (ch, up) in enumerate(zip(c["chans"], c["ups"])):
        hh, ww = (c["h"] // 2, c["w"] // 2) if up else (c["h"], c["w"])
        srcs.append((ops.as_nhwc(_t(syn.hash_normalish((c["n"], ch, hh, ww), c["seed"] + 10 + i))), up))
    ho = (c["h"] + 2 * (c["k"] // 2) - c["k"]) // c["st"] + 1
    wo = (c["w"] + 2 * (c["k"] // 2) - c["k"]) // c["st"] + 1
    res = ops.as_nhwc(_t(syn.hash_normalish((c["n"], c["cout"], ho, wo), c["seed"] + 20))) if c["res"] else None
    want = ops.conv2d(srcs, conv, act=c["act"], residual=res, impl="simple")
    got = ops.conv2d(srcs, conv, act=c["act"], residual=res)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (c["n"], c["cout"], ho, wo)
    assert (got - want).abs().max().item() < 4e-6 * max(want.abs().max().item(), 5.0)
    assert got.abs().max().item() > 0.05
    assert torch.equal(ops.conv2d(srcs, conv, act=c["act"], residual=res), got)


@pytest.mark.parametrize("shape", [(1, 64, 64, 120, 160, 1), (2, 128, 64, 60, 80, 2), (1, 64, 32, 37, 53, 1), (1, 96, 64, 24, 40, 3)])
def test_split_precision_winograd_conv_matches_the_fp32_kernels(shape):
    """Opt-in split-precision Winograd conv (csrc/conv_wino_split.hip: fp16 hi/lo operand pairs on the fp16 matrix pipe,
    fp32 accumulation) against the exact-fp32 Winograd kernel and the direct general-shape kernel on the same inputs:
    multi-source concat with a nearest x2 upsampled source, bias, residual, LeakyReLU / ELU, ragged extents, batch 2."""
    import gpu_util as gu
    import torch.nn as nn

    from doubletake_amd.modules import conv_ops as ops

    n, cin, cout, h, w, nsrc = shape
    dev = gu.dev()
    conv = nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    gu.set_formula_weights(conv, 31 + cin)
    cs = [cin] if nsrc == 1 else ([cin // 2, cin // 2] if nsrc == 2 else [cin // 3, cin // 3, cin // 3])
    up_first = nsrc >= 2 and h % 2 == 0 and w % 2 == 0
    srcs = []
    for i, c in enumerate(cs):
        hh, ww = (h // 2, w // 2) if (i == 0 and up_first) else (h, w)
        x = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, c, hh, ww), 50 + i)).to(dev))
        srcs.append((x, i == 0 and up_first))
    res = ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, cout, h, w), 77)).to(dev))
    for act in (ops.ACT_LRELU02, ops.ACT_ELU):
        want = ops.conv2d(srcs, conv, act=act, residual=res, impl="wino")
        simple = ops.conv2d(srcs, conv, act=act, residual=res, impl="simple")
        prev = ops.CONV_PRECISION, ops.SPLIT_MIN_BLOCKS
        ops.CONV_PRECISION, ops.SPLIT_MIN_BLOCKS = "split16", 1
        try:
            got = ops.conv2d(srcs, conv, act=act, residual=res, impl="wino")
        finally:
            ops.CONV_PRECISION, ops.SPLIT_MIN_BLOCKS = prev
        torch.cuda.synchronize()
        scale = max(float(simple.abs().max()), 1.0)
        assert float((got - want).abs().max()) < 3e-6 * scale, (act, float((got - want).abs().max()))
        assert float((got - simple).abs().max()) < 6e-6 * scale
        assert float((want - simple).abs().max()) < 6e-6 * scale


@pytest.mark.parametrize("mask", [11, 31])
def test_throughput_plan_objective_computes_the_same_graph(mask):
    """dt_conv_set_plan_objective (round 5): with several keyframes in flight the launchers give up the in-workgroup /
    cross-workgroup K splits that only buy single-launch latency.  Same products, possibly another fp32 summation order:
    the small-model graph at full bench size must agree with the latency plan to rounding and both with the reference golden
    at the small size; the setter returns the mask in force and 0 restores the default."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules.networks import CVEncoder
    from doubletake_amd.modules.networks_fast import SkipDecoderRegression

    g = load_golden("networks.npz")
    enc = [64, 64, 128, 256, 512]
    cve = CVEncoder(D, enc[1:], [64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(cve, 1234)
    dec = SkipDecoderRegression([enc[0], 64, 128, 256, 384]).to(gu.dev())
    gu.set_formula_weights(dec, 2345)

    def run(h0, w0, seed):
        vol = _t(syn.hash_normalish((1, D, h0, w0), seed))
        feats = [_t(f) for f in syn.prior_pyramid(1, enc, 2 * h0, 2 * w0, 555)]
        outs = cve(vol, feats[1:])
        dout = dec([feats[0]] + outs)
        torch.cuda.synchronize()
        return outs, dout

    try:
        assert ops.set_plan_objective(ops.PLAN_LATENCY) == 0
        lat_small, lat_big = run(H0, W0, 4321), run(120, 160, 77)
        assert ops.set_plan_objective(mask) == mask
        thr_small, thr_big = run(H0, W0, 4321), run(120, 160, 77)
    finally:
        assert ops.set_plan_objective(ops.PLAN_LATENCY) == 0
    for i, o in enumerate(thr_small[0]):
        np.testing.assert_allclose(_np(o), g[f"cve_small_out{i}"], atol=1e-4, rtol=0)
    for k, v in thr_small[1].items():
        np.testing.assert_allclose(_np(v), g[f"skip_{k}"], atol=3e-4, rtol=0)
    for (a, b) in list(zip(lat_big[0], thr_big[0])) + [(lat_big[1][k], thr_big[1][k]) for k in lat_big[1]]:
        scale = max(a.abs().max().item(), 1.0)
        assert (a - b).abs().max().item() < 2e-5 * scale
    assert ops.PLAN_THROUGHPUT == 11 and ops.WINO_MIN_BLOCKS == 96  # (restored with the latency plan)


@pytest.mark.parametrize("h0,w0,with_depth", [(120, 160, True), (96, 128, False), (16, 24, True)])
def test_coarse_heads_inside_the_conv_grid_are_bit_identical(h0, w0, with_depth):
    """Round 5 (dt_conv2d_wino_heads_f32): SkipDecoderRegression's heads of scales 3, 2, 1 run as extra workgroups (behind the conv blocks) of the
    last block's first 240x320 convolution instead of as a launch of their own.  Same bodies: every output of the decoder must
    equal the two-launch path bit for bit, and the step must need one launch less -- at bench size and at the cfg4 size; on a
    small map (the convolution does not fill the chip) the fused entry falls back to the two launches."""
    import gpu_util as gu
    from doubletake_amd import _abi
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules.networks_fast import SkipDecoderRegression

    enc = [64, 64, 128, 256, 384]
    dec = SkipDecoderRegression(enc).to(gu.dev())
    gu.set_formula_weights(dec, 2345)
    feats = [_t(syn.hash_normalish((1, c, (2 * h0) >> i, (2 * w0) >> i), 70 + i)) for i, c in enumerate(enc)]
    L = _abi.lib()

    def run(flag):
        prev = ops.HEADS_IN_CONV
        ops.HEADS_IN_CONV = flag
        try:
            dec(feats, with_depth=with_depth)  # (weight packs)
            c0 = int(L.dt_kernel_launch_count())
            out = dec(feats, with_depth=with_depth)
            torch.cuda.synchronize()
            return out, int(L.dt_kernel_launch_count()) - c0
        finally:
            ops.HEADS_IN_CONV = prev

    fused, n_fused = run(True)
    plain, n_plain = run(False)
    assert set(fused) == set(plain)
    for k in plain:
        assert torch.equal(fused[k], plain[k]), k
    assert float(fused["log_depth_pred_s3_b1hw"].abs().max()) > 1e-3
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    blocks = ((2 * h0 + 7) // 8) * ((2 * w0 + 15) // 16) * 2
    assert n_fused == n_plain - (1 if blocks >= 2 * cus else 0)


@pytest.mark.parametrize("shape", [
    (1, 16, 32, 13, 21, "zeros", 1, False),      # ragged extent (one partial 16x16 block in each direction), one source
    (2, 24, 64, 16, 32, "zeros", 2, True),       # batch 2, two sources, the second nearest-upsampled, residual + bias
    (1, 64, 32, 24, 40, "replicate", 1, False),  # replicate padding
    (1, 64, 64, 120, 160, "zeros", 1, True),     # 8 x 10 blocks x 2 channel blocks, last block row half outside the image
    (1, 192, 64, 48, 64, "zeros", 3, True),      # three sources (UNet++ node input), the first nearest-upsampled
    (2, 128, 128, 96, 128, "zeros", 1, False),   # cfg3-like level: 384 blocks, XCD-contiguous order
])
def test_winograd_f4x4_conv_vs_direct(shape):
    """conv_wino4_kernel (F(4x4,3x3) on v_mfma_f32_16x16x4_f32, csrc/conv_wino4.hip) against the direct one-thread-per-output
    conv and against the F(2x2) kernel on the same inputs: fused concat / nearest x2 / padding / bias / residual / activation.
    Tolerance: F(4x4) amplifies fp32 rounding (numpy model of the same arithmetic: 3e-5 worst case on O(1) outputs)."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    n, cin, cout, h, w, pad_mode, nsrc, with_res = shape
    dev = gu.dev()
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, padding_mode=pad_mode).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(syn.hash_normalish(tuple(conv.weight.shape), 5) * (1.0 / np.sqrt(9.0 * cin))))
        conv.bias.copy_(torch.from_numpy(syn.hash_normalish((cout,), 6) * 0.1))
    mk = lambda c, hh, ww, seed: ops.as_nhwc(torch.from_numpy(syn.hash_normalish((n, c, hh, ww), seed)).to(dev))
    if nsrc == 1:
        srcs = [(mk(cin, h, w, 1), False)]
    elif nsrc == 2:
        srcs = [(mk(cin - 8, h, w, 1), False), (mk(8, h // 2, w // 2, 2), True)]
    else:
        srcs = [(mk(cin // 3, h // 2, w // 2, 1), True), (mk(cin // 3, h, w, 2), False), (mk(cin // 3, h, w, 3), False)]
    res = mk(cout, h, w, 4) if with_res else None
    prev = ops.WINO4_MIN_BLOCKS
    for act in (ops.ACT_LRELU02, ops.ACT_ELU, ops.ACT_NONE):
        want = ops.conv2d(srcs, conv, act=act, residual=res, impl="simple")
        ops.WINO4_MIN_BLOCKS = 0
        try:
            f2 = ops.conv2d(srcs, conv, act=act, residual=res, impl="wino")
            ops.WINO4_MIN_BLOCKS = 1
            got = ops.conv2d(srcs, conv, act=act, residual=res, impl="wino")
            again = ops.conv2d(srcs, conv, act=act, residual=res, impl="wino")
        finally:
            ops.WINO4_MIN_BLOCKS = prev
        torch.cuda.synchronize()
        assert got.shape == want.shape and torch.equal(got, again)
        scale = max(float(want.abs().max()), 1.0)
        err = float((got - want).abs().max())
        assert err < 4e-5 * scale, (act, err, scale)
        assert float((f2 - want).abs().max()) < 4e-6 * max(scale, 5.0)
        assert not torch.equal(got, f2)  # (the F(4x4) kernel really ran: its rounding differs from F(2x2)'s)
