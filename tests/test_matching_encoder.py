"""ResnetMatchingEncoder: state-dict layout (CPU) and GPU parity against the torch CPU reference."""
import numpy as np
import pytest
import torch

from doubletake_amd.utils import synthetic as syn

REF_KEYS_TAIL = ["net.5.weight", "net.5.bias", "net.8.weight", "net.8.bias"]


def _block_keys(p):
    out = []
    for c, b in (("conv1", "bn1"), ("conv2", "bn2")):
        out.append(f"{p}.{c}.weight")
        out += [f"{p}.{b}.{n}" for n in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    return out


def _expected_keys(antialiased):
    keys = ["net.0.weight"] + [f"net.1.{n}" for n in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    if antialiased:
        keys.append("net.3.1.filt")
    return keys + _block_keys("net.4.0") + _block_keys("net.4.1") + REF_KEYS_TAIL


def _fill(m, seed):
    """Deterministic weights; BatchNorm statistics away from the identity so that folding is exercised."""
    from doubletake_amd.utils.synthetic import formula_params

    with torch.no_grad():
        named = [(n, p) for n, p in m.named_parameters()]
        for (n, p), a in zip(named, formula_params([tuple(p.shape) for _, p in named], seed)):
            p.copy_(torch.from_numpy(a))
        j = 0
        for n, b in m.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.from_numpy(syn.hash_normalish(tuple(b.shape), seed + 100 + j) * 0.2))
            elif n.endswith("running_var"):
                b.copy_(torch.from_numpy(0.5 + syn.hash_u01(tuple(b.shape), seed + 200 + j).astype(np.float32)))
            j += 1
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(torch.from_numpy(0.8 + 0.4 * syn.hash_u01((mod.num_features,), seed + 300 + j).astype(np.float32)))
                mod.bias.copy_(torch.from_numpy(0.1 * syn.hash_normalish((mod.num_features,), seed + 400 + j)))
                j += 1
    return m


@pytest.mark.parametrize("antialiased", [True, False])
def test_state_dict_layout_matches_reference_children(antialiased):
    from doubletake_amd.modules.networks import ResnetMatchingEncoder

    m = ResnetMatchingEncoder(18, 16, pretrained=False, antialiased=antialiased)
    sd = m.state_dict()
    assert list(sd.keys()) == _expected_keys(antialiased)
    assert tuple(sd["net.0.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["net.5.weight"].shape) == (128, 64, 1, 1)
    assert tuple(sd["net.8.weight"].shape) == (16, 128, 3, 3)
    assert m.net[8].padding_mode == "replicate"
    if antialiased:
        f = sd["net.3.1.filt"]
        assert tuple(f.shape) == (64, 1, 4, 4) and abs(float(f[0].sum()) - 1.0) < 1e-6
        assert torch.allclose(f[0, 0, 0] * 64, torch.tensor([1.0, 3.0, 3.0, 1.0]))


class _TorchBlock(torch.nn.Module):
    """BasicBlock(64, 64) as torchvision writes it, for the module-vs-functional oracle check below."""

    def __init__(self):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False)
        self.bn1 = torch.nn.BatchNorm2d(64)
        self.conv2 = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False)
        self.bn2 = torch.nn.BatchNorm2d(64)

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        return torch.relu(self.bn2(self.conv2(y)) + x)


def test_oracle_against_torch_modules():
    """The functional oracle equals an nn.Sequential of torch layers (torchvision-style stem) that loads
    the same state dict."""
    from doubletake_amd.modules.networks import ResnetMatchingEncoder
    from oracle import matching_encoder_ref as mref

    nn = torch.nn
    m = _fill(ResnetMatchingEncoder(18, 16, pretrained=False, antialiased=False), 5)
    net = nn.Sequential(
        nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1),
        nn.Sequential(_TorchBlock(), _TorchBlock()), nn.Conv2d(64, 128, 1), nn.InstanceNorm2d(128), nn.LeakyReLU(0.2),
        nn.Conv2d(128, 16, 3, padding=1, padding_mode="replicate"), nn.InstanceNorm2d(16)).eval()
    net.load_state_dict({k[len("net."):]: v for k, v in m.state_dict().items()})
    img = syn.hash_normalish((2, 3, 32, 48), 9)
    with torch.no_grad():
        want = net(torch.from_numpy(img)).numpy()
    got = mref.matching_encoder(img, m.state_dict(), antialiased=False)
    assert np.abs(got - want).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("antialiased,shape", [(True, (2, 3, 64, 96)), (False, (2, 3, 64, 96)), (True, (1, 3, 52, 76))])
def test_encoder_vs_cpu_reference(antialiased, shape):
    import gpu_util as gu
    from doubletake_amd.modules.networks import ResnetMatchingEncoder
    from oracle import matching_encoder_ref as mref

    m = _fill(ResnetMatchingEncoder(18, 16, pretrained=False, antialiased=antialiased), 7)
    img = syn.hash_normalish(shape, 11)
    want = mref.matching_encoder(img, m.state_dict(), antialiased=antialiased)
    md = m.to(gu.dev())
    got = md(torch.from_numpy(img).to(gu.dev()))
    assert tuple(got.shape) == want.shape and got.is_contiguous()
    # instance-normalised outputs are O(1); fp32 MFMA sums vs ATen's CPU convs
    assert np.abs(got.cpu().numpy() - want).max() < 3e-4
    got_simple = md(torch.from_numpy(img).to(gu.dev()), _impl="simple")
    assert np.abs(got_simple.cpu().numpy() - want).max() < 3e-4
    nhwc = md(torch.from_numpy(img).to(gu.dev()), channels_last_output=True)
    assert torch.equal(nhwc.contiguous(), got)


@pytest.mark.gpu
def test_encoder_ops_vs_torch():
    import torch.nn.functional as F

    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules import matching_encoder as me

    x = torch.from_numpy(syn.hash_normalish((2, 64, 23, 31), 3))
    xd = ops.as_nhwc(x.to(gu.dev()))
    for k, s, p in ((3, 2, 1), (2, 1, 0)):
        assert torch.equal(me.maxpool(xd, k, s, p).cpu().contiguous(), F.max_pool2d(x, k, s, p))
    blur = me.BlurPool(64)
    want = F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), blur.filt, stride=2, groups=64)
    assert (me.blurpool(xd, blur).cpu() - want).abs().max() < 1e-6
    want2 = F.conv2d(F.pad(F.max_pool2d(x, 2, 1), (1, 2, 1, 2), mode="reflect"), blur.filt, stride=2, groups=64)
    got2 = me.maxblur(xd, blur).cpu()
    assert got2.shape == want2.shape and (got2 - want2).abs().max() < 1e-6
    for c, act in ((64, ops.ACT_LRELU02), (16, ops.ACT_NONE)):
        want = F.instance_norm(x[:, :c], eps=1e-5)
        if act:
            want = F.leaky_relu(want, 0.2)
        got = me.instance_norm(xd, c, 1e-5, act=act, out_nchw=True)
        assert (got.cpu() - want).abs().max() < 2e-5
        got2 = me.instance_norm(xd, c, 1e-5, act=act, out_nchw=False)
        assert torch.equal(got2.contiguous(), got)


@pytest.mark.gpu
def test_replicate_padding_and_relu_conv():
    import torch.nn.functional as F

    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops

    conv = torch.nn.Conv2d(16, 32, 3, padding=1, padding_mode="replicate")
    x = torch.from_numpy(syn.hash_normalish((1, 16, 13, 21), 4))
    with torch.no_grad():
        want = F.relu(conv(x))
    cd = conv.to(gu.dev())
    for impl in ("mfma", "simple"):
        got = ops.conv2d([(ops.as_nhwc(x.to(gu.dev())), False)], cd, act=ops.ACT_RELU, impl=impl)
        assert (got.cpu() - want).abs().max() < 2e-5, impl


@pytest.mark.gpu
def test_model_matching_feats_and_cross_frame_cache():
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint

    model = DepthModelCVHint(64, 96, depth_decoder_name="skip", matching_num_depth_bins=8, model_num_views=4)
    _fill(model.matching_model, 21)
    model = model.to(gu.dev())
    b, k = 2, 3
    cur = torch.from_numpy(syn.hash_normalish((b, 3, 64, 96), 31)).to(gu.dev())
    src = torch.from_numpy(syn.hash_normalish((b, k, 3, 64, 96), 32)).to(gu.dev())
    m_cur, m_src = model.compute_matching_feats(cur, src)
    assert tuple(m_cur.shape) == (b, 16, 16, 24) and tuple(m_src.shape) == (b, k, 16, 16, 24)
    u_cur, u_src = model.compute_matching_feats(cur, src, unbatched_matching_encoder_forward=True)
    assert (u_cur - m_cur).abs().max() < 1e-4 and (u_src - m_src).abs().max() < 1e-4
    # cached: ids laid out like src_data["frame_id_string"] (K lists of b strings)
    cur_ids = ["f10", "f20"]
    src_ids = [["f9", "f19"], ["f8", "f18"], ["f7", "f10"]]  # f10 is also a source of the second element
    src2 = src.clone()
    src2[1, 2] = cur[0]
    c_cur, c_src = model.compute_matching_feats(cur, src2, cur_ids=cur_ids, src_ids=src_ids)
    cache = model.matching_feature_cache
    assert len(cache) == 7 and cache.misses == 7
    ref_cur, ref_src = model.compute_matching_feats(cur, src2)
    assert (c_cur - ref_cur).abs().max() < 1e-4 and (c_src - ref_src).abs().max() < 1e-4
    assert torch.equal(c_src[1, 2], c_cur[0])
    # next tuple: one new frame, everything else served from HBM
    cur_b = torch.from_numpy(syn.hash_normalish((b, 3, 64, 96), 33)).to(gu.dev())
    model.compute_matching_feats(cur_b, src2, cur_ids=["f11", "f21"], src_ids=src_ids)
    assert cache.misses == 9 and len(cache) == 9
    cache.capacity = 4
    cache.put("x", c_cur[0])
    assert len(cache) == 4 and "x" in cache
    # a capacity below one call's b * (1 + K) frames: the call still gets every map (taken out before the insertions)
    cache.clear()
    cache.capacity = 3
    s_cur, s_src = model.compute_matching_feats(cur, src2, cur_ids=cur_ids, src_ids=src_ids)
    assert torch.equal(s_cur, c_cur) and torch.equal(s_src, c_src) and len(cache) == 3
    t_cur, t_src = model.compute_matching_feats(cur, src2, cur_ids=cur_ids, src_ids=src_ids)   # partly cached now
    assert (t_cur - c_cur).abs().max() < 1e-4 and (t_src - c_src).abs().max() < 1e-4


@pytest.mark.gpu
def test_feature_cache_is_keyed_by_scan_and_weights():
    """Frame ids repeat across scans and entries die with the weights that produced them."""
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint

    model = DepthModelCVHint(64, 96, depth_decoder_name="skip", matching_num_depth_bins=8, model_num_views=3)
    _fill(model.matching_model, 21)
    model = model.to(gu.dev())
    img = lambda s: torch.from_numpy(syn.hash_normalish((1, 3, 64, 96), s)).to(gu.dev())
    src = lambda s: torch.from_numpy(syn.hash_normalish((1, 2, 3, 64, 96), s)).to(gu.dev())
    ids = dict(cur_ids=["000012"], src_ids=[["000010"], ["000008"]])
    a_cur, _ = model.compute_matching_feats(img(1), src(2), scan_ids="scene0707_00", **ids)
    # same frame-id strings, different scan and different pixels: must NOT be served from the first scan's entries
    b_cur, _ = model.compute_matching_feats(img(3), src(4), scan_ids="scene0708_00", **ids)
    want, _ = model.compute_matching_feats(img(3), src(4))
    assert (b_cur - want).abs().max() < 1e-4 and (b_cur - a_cur).abs().max() > 1e-2
    cache = model.matching_feature_cache
    assert len(cache) == 6
    # a weight update invalidates everything
    with torch.no_grad():
        next(model.matching_model.parameters()).mul_(1.5)
    c_cur, _ = model.compute_matching_feats(img(1), src(2), scan_ids="scene0707_00", **ids)
    want, _ = model.compute_matching_feats(img(1), src(2))
    assert len(cache) == 3 and (c_cur - want).abs().max() < 1e-4 and (c_cur - a_cur).abs().max() > 1e-3


def _blur_handcases():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "blurpool_handcases.json")))


_SCALES = (1.0, 2.0, 4.0, 0.5)  # four channels (the kernels work on channel quads); positive, so max commutes


@pytest.mark.parametrize("case", _blur_handcases(), ids=lambda c: c["name"])
def test_antialiased_maxpool_restatement_reproduces_hand_derived_cases(case):
    """The torch restatement the oracle uses for `antialiased_cnns` maxpool = Sequential(MaxPool2d(2, 1), BlurPool(64, 4, 2))
    against answers derived from the package's published rules in exact arithmetic (tests/golden/make_blurpool_handcases.py):
    EXACT equality -- every expected value is a small multiple of 1/64."""
    import torch.nn.functional as F

    from doubletake_amd.modules import matching_encoder as me

    x = torch.tensor(case["x"], dtype=torch.float32)[None, None] * torch.tensor(_SCALES).view(1, 4, 1, 1)
    blur = me.BlurPool(4)
    a = torch.tensor([1.0, 3.0, 3.0, 1.0])
    assert torch.equal(blur.filt, (a[:, None] * a[None, :] / 64.0)[None, None].repeat(4, 1, 1, 1))
    m = F.max_pool2d(x, kernel_size=2, stride=1)
    assert torch.equal(m[0, 0], torch.tensor(case["maxpool2_s1"], dtype=torch.float32))
    got = F.conv2d(F.pad(m, (1, 2, 1, 2), mode="reflect"), blur.filt, stride=2, groups=4)
    for ch, sc in enumerate(_SCALES):
        assert torch.equal(got[0, ch], torch.tensor(case["maxblur"], dtype=torch.float32) * sc), (case["name"], ch)
    got_b = F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), blur.filt, stride=2, groups=4)
    assert torch.equal(got_b[0, 0], torch.tensor(case["blur"], dtype=torch.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", _blur_handcases(), ids=lambda c: c["name"])
def test_antialiased_maxpool_kernels_reproduce_hand_derived_cases(case):
    """dt_maxblur_f32 (fused), dt_maxpool_f32 + dt_blurpool4_s2_f32 (separate) against the same hand-derived answers, exactly."""
    import gpu_util as gu
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules import matching_encoder as me

    x = torch.tensor(case["x"], dtype=torch.float32)[None, None] * torch.tensor(_SCALES).view(1, 4, 1, 1)
    xd = ops.as_nhwc(x.to(gu.dev()))
    blur = me.BlurPool(4).to(gu.dev())
    want = torch.tensor(case["maxblur"], dtype=torch.float32)
    fused = me.maxblur(xd, blur).cpu()
    two = me.blurpool(me.maxpool(xd, 2, 1, 0), blur).cpu()
    alone = me.blurpool(xd, blur).cpu()
    for ch, sc in enumerate(_SCALES):
        assert torch.equal(fused[0, ch], want * sc) and torch.equal(two[0, ch], want * sc), (case["name"], ch)
        assert torch.equal(alone[0, ch], torch.tensor(case["blur"], dtype=torch.float32) * sc)


# ---- second, independent formulation of the anti-aliased max-pool -----------------------------------
def _maxblur_separable_numpy(x):
    """MaxPool2d(2, stride 1) + BlurPool(filt 4, stride 2) written WITHOUT convolutions or torch padding ops: explicit
    reflected index arithmetic and two separable 1-D binomial passes [1, 3, 3, 1] / 8 (rows, then columns) in float64.
    Reflection (no edge repeat) of index i on [0, n): -1 -> 1, n -> n - 2, n + 1 -> n - 3 (ReflectionPad2d((1, 2, 1, 2)))."""
    x = np.asarray(x, dtype=np.float64)
    m = np.maximum(np.maximum(x[..., :-1, :-1], x[..., :-1, 1:]), np.maximum(x[..., 1:, :-1], x[..., 1:, 1:]))
    taps = np.array([1.0, 3.0, 3.0, 1.0]) / 8.0

    def reflect(i, n):
        return -i if i < 0 else (2 * (n - 1) - i if i >= n else i)

    def pass_1d(a, axis):
        n = a.shape[axis]
        n_out = (n + 3 - 4) // 2 + 1
        out = []
        for o in range(n_out):
            acc = 0.0
            for t_i, wgt in enumerate(taps):
                acc = acc + wgt * np.take(a, reflect(2 * o + t_i - 1, n), axis=axis)
            out.append(acc)
        return np.stack(out, axis=axis)

    return pass_1d(pass_1d(m, -2), -1)


def test_separable_formulation_agrees_with_the_restatement_and_the_hand_cases():
    """Three independent statements of the package's published rules must agree: the exact hand cases, the torch
    restatement the oracle uses (2-D depthwise correlation after ReflectionPad2d), and the separable index-arithmetic
    form above -- on the hand cases exactly and on random maps (odd and even sizes, incl. the smallest legal one) to
    float rounding."""
    import torch.nn.functional as F

    from doubletake_amd.modules import matching_encoder as me

    for case in _blur_handcases():
        x = np.asarray(case["x"], dtype=np.float64)
        np.testing.assert_allclose(_maxblur_separable_numpy(x[None, None])[0, 0], np.asarray(case["maxblur"]), rtol=0, atol=1e-12)
    rng = np.random.default_rng(5)
    for (h, w) in ((5, 7), (8, 8), (9, 12), (16, 21), (4, 5)):
        x = rng.standard_normal((2, 3, h, w)).astype(np.float32)
        blur = me.BlurPool(3)
        xt = torch.from_numpy(x)
        want = F.conv2d(F.pad(F.max_pool2d(xt, 2, 1), (1, 2, 1, 2), mode="reflect"), blur.filt, stride=2, groups=3).numpy()
        got = _maxblur_separable_numpy(x)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_maxblur_kernels_agree_with_the_separable_formulation_on_random_maps():
    """The HIP kernels (fused dt_maxblur_f32; dt_maxpool_f32 + dt_blurpool4_s2_f32) against the separable float64
    formulation on random maps, not only on the hand cases."""
    import gpu_util as gu
    from doubletake_amd.modules import matching_encoder as me

    rng = np.random.default_rng(6)
    for (n, c, h, w) in ((1, 64, 120, 160), (2, 8, 9, 12), (1, 4, 17, 5), (3, 16, 32, 33)):
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        xd = torch.from_numpy(x).to(gu.dev()).contiguous(memory_format=torch.channels_last)
        blur = me.BlurPool(c).to(gu.dev())
        want = _maxblur_separable_numpy(x)
        fused = me.maxblur(xd, blur).cpu().numpy()
        two = me.blurpool(me.maxpool(xd, 2, 1, 0), blur).cpu().numpy()
        assert fused.shape == want.shape
        np.testing.assert_allclose(fused, want, rtol=0, atol=3e-6)
        np.testing.assert_allclose(two, want, rtol=0, atol=3e-6)
