"""ORACLE (test infrastructure, NOT product code) -- numpy float32 restatement of the
cost-volume encoder / depth decoders of the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity pin: tests/golden/networks.npz (captured from the imported reference by
tests/golden/make_golden.py), checked in tests/test_oracle_networks.py.

Weights are passed as dicts keyed by the reference's state-dict names (e.g.
"convs.ds_conv_0.conv1.weight"), so the structure of each function mirrors the module it
restates (paths relative to /root/reference/src/doubletake/).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def conv2d(x, W, b=None, stride=1, pad=None):
    """nn.Conv2d (cross-correlation, zero padding).  x [n,c,h,w], W [co,ci,k,k]."""
    n, c, h, w = x.shape
    co, ci, k, _ = W.shape
    assert ci == c
    if pad is None:
        pad = k // 2
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    xp = np.zeros((n, c, h + 2 * pad, w + 2 * pad), dtype=F32)
    xp[:, :, pad:pad + h, pad:pad + w] = x
    cols = np.empty((n, c, k, k, ho, wo), dtype=F32)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, ky, kx] = xp[:, :, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride]
    cols = cols.reshape(n, c * k * k, ho * wo)
    out = np.matmul(W.reshape(co, -1).astype(F32)[None], cols)
    if b is not None:
        out = out + b.reshape(1, co, 1).astype(F32)
    return out.reshape(n, co, ho, wo).astype(F32)


def lrelu(x, slope=0.2):
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)).astype(F32)).astype(F32)


def upsample_nearest2(x):
    """F.interpolate(scale_factor=2, mode='nearest') (modules/networks_fast.py:36)."""
    return x.repeat(2, axis=2).repeat(2, axis=3)


def upsample_bilinear2(x):
    """utils/generic_utils.py:95-104: F.interpolate(scale_factor=2, bilinear, align_corners=False)."""
    n, c, h, w = x.shape

    def idx(size):
        dst = np.arange(2 * size, dtype=F32)
        src = np.maximum((dst + F32(0.5)) * F32(0.5) - F32(0.5), F32(0.0))
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, size - 1)
        l1 = (src - i0.astype(F32)).astype(F32)
        return i0, i1, (F32(1.0) - l1).astype(F32), l1

    y0, y1, wy0, wy1 = idx(h)
    x0, x1, wx0, wx1 = idx(w)
    rows = x[:, :, y0] * wy0[None, None, :, None] + x[:, :, y1] * wy1[None, None, :, None]
    out = rows[:, :, :, x0] * wx0 + rows[:, :, :, x1] * wx1
    return out.astype(F32)


def _p(params, prefix):
    return {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}


def basic_block(x, p, stride=1):
    """modules/layers.py:77-94 with norm_layer=nn.Identity (bias on, LeakyReLU 0.2, no BN)."""
    out = lrelu(conv2d(x, p["conv1.weight"], p["conv1.bias"], stride=stride))
    out = conv2d(out, p["conv2.weight"], p["conv2.bias"])
    if "downsample.0.weight" in p:
        identity = conv2d(x, p["downsample.0.weight"], p["downsample.0.bias"], stride=stride)
    else:
        identity = x
    return lrelu(out + identity)


def cv_encoder(x, img_feats, params):
    """CVEncoder.forward, modules/networks.py:110-117 (construction :89-108)."""
    outs = []
    nblocks = len(img_feats)
    for i in range(nblocks):
        x = basic_block(x, _p(params, f"convs.ds_conv_{i}."), stride=1 if i == 0 else 2)
        x = np.concatenate([x, img_feats[i]], axis=1)
        x = basic_block(x, _p(params, f"convs.conv_{i}.0."))
        x = basic_block(x, _p(params, f"convs.conv_{i}.1."))
        outs.append(x)
    return outs


def conv_block(x, p):
    """ConvBlock, modules/networks_fast.py:6-24: (conv3x3 + ELU) x 2."""
    x = elu(conv2d(x, p["conv1.weight"], p["conv1.bias"]))
    return elu(conv2d(x, p["conv2.weight"], p["conv2.bias"]))


def skip_decoder_regression(features, params):
    """SkipDecoderRegression.forward, modules/networks_fast.py:79-141."""
    out = {}
    x = features[-1]
    for bi, scale in ((1, 3), (2, 2), (3, 1), (4, 0)):
        p = _p(params, f"block{bi}.")
        x = conv_block(x, _p(p, "pre_concat_conv."))
        x = upsample_nearest2(x)
        x = np.concatenate([x, features[-1 - bi]], axis=1)
        x = conv_block(x, _p(p, "post_concat_conv."))
        out[f"feature_s{scale}_b1hw"] = x
        hp = _p(params, f"out{bi}.")
        y = elu(conv2d(x, hp["0.weight"], hp["0.bias"]))
        y = elu(conv2d(y, hp["2.weight"], hp["2.bias"]))
        out[f"log_depth_pred_s{scale}_b1hw"] = conv2d(y, hp["4.weight"], hp["4.bias"])
    return out


def _double_basic_block(x, p):
    """modules/networks.py:13-17: Sequential(BasicBlock '0', BasicBlock 'conv_0')."""
    x = basic_block(x, _p(p, "0."))
    return basic_block(x, _p(p, "conv_0."))


def depth_decoder_pp(input_features, params):
    """DepthDecoderPP.forward, modules/networks.py:65-85 (UNet++ grid; head per scale :60-63)."""
    prev = list(input_features)
    outputs = []
    depth = {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            ins = [basic_block(prev[i], _p(params, f"convs.right_conv_{i}{j - 1}."))]
            ins.append(upsample_bilinear2(basic_block(prev[i + 1], _p(params, f"convs.diag_conv_{i + 1}{j - 1}."))))
            if i + j != 4:
                ins.append(upsample_bilinear2(basic_block(outputs[-1], _p(params, f"convs.up_conv_{i + 1}{j}."))))
            out = _double_basic_block(np.concatenate(ins, axis=1), _p(params, f"convs.in_conv_{i}{j}."))
            outputs.append(out)
            hp = _p(params, f"convs.output_{i}.")
            y = out
            if i != 0:
                y = basic_block(y, _p(hp, "0."))
            depth[f"log_depth_pred_s{i}_b1hw"] = conv2d(y, hp["1.weight"], hp["1.bias"])
        prev = outputs[::-1]
    return depth
