"""ResnetMatchingEncoder (reference modules/networks.py:138-189) on the HIP conv primitive.

    image [B,3,H,W] -> conv1 7x7/2 + bn1 + relu -> maxpool -> layer1 (2 BasicBlocks, 64 ch)
                    -> 1x1 conv 64->128 -> InstanceNorm -> LeakyReLU(0.2)
                    -> 3x3 replicate-pad conv 128->C -> InstanceNorm          [B,C,H/4,W/4]

The reference takes the first five children of a torchvision / antialiased_cnns ResNet.  Neither
package is installed here, so the stem is rebuilt from torch.nn layers with the SAME child names
and parameter shapes (state dicts of the reference load unchanged):

    net.0  conv1   Conv2d(3,64,7,stride 2,pad 3,bias=False)
    net.1  bn1     BatchNorm2d(64)          (inference: folded into net.0's weights on the host)
    net.2  relu
    net.3  maxpool torchvision: MaxPool2d(3,2,1);  anti-aliased: Sequential(MaxPool2d(2,1), BlurPool(64, filt 4, stride 2))
    net.4  layer1  Sequential(BasicBlock(64,64), BasicBlock(64,64))   conv3x3-bn-relu-conv3x3-bn (+x) relu
    net.5..9       as written in the reference

The anti-aliased `maxpool` (the only non-torch layer) follows antialiased_cnns' blurpool.py / resnet.py and is pinned by
the hand-derived exact cases of tests/golden/make_blurpool_handcases.py; the torchvision variant and everything from
net.5 on are plain torch.nn semantics.

Eval-mode only (BatchNorm uses running statistics), like every caller in the reference's test scripts.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _abi
from . import conv_ops as ops


class BlurPool(nn.Module):
    """Parameter holder for the anti-aliasing blur: buffer ``filt`` [C,1,4,4] (binomial, sums to 1)."""

    def __init__(self, channels, filt_size=4, stride=2):
        super().__init__()
        if filt_size != 4 or stride != 2:
            raise NotImplementedError("only BlurPool(filt_size=4, stride=2) is used by the matching encoder")
        a = np.array([1.0, 3.0, 3.0, 1.0])
        f = torch.tensor(a[:, None] * a[None, :], dtype=torch.float32)
        f = f / f.sum()
        self.channels = channels
        self.register_buffer("filt", f[None, None].repeat(channels, 1, 1, 1))


class ResnetBasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock(64, 64) without downsample (parameter holder)."""

    def __init__(self, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)


def _fold_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d, device, reshape_1x1_pad_to=None):
    """Conv2d holder with BatchNorm (running stats) folded in: w' = w * g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps).
    Cached on the conv module, keyed on the versions of every tensor involved."""
    tensors = [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = (str(device), reshape_1x1_pad_to) + tuple((t.data_ptr(), t._version) for t in tensors)
    cache = conv.__dict__.setdefault("_dt_folded", {})
    hit = cache.get(reshape_1x1_pad_to)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        scale = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
        w = (conv.weight.double() * scale.view(-1, 1, 1, 1)).float()
        b = (bn.bias.double() - bn.running_mean.double() * scale).float()
        if conv.bias is not None:
            b = b + (conv.bias.double() * scale).float()
        if reshape_1x1_pad_to is not None:  # stem: [co,3,7,7] -> [co,152,1,1] in im2col column order
            co = w.shape[0]
            flat = w.reshape(co, -1)
            w = torch.cat([flat, flat.new_zeros(co, reshape_1x1_pad_to - flat.shape[1])], 1).view(co, -1, 1, 1)
            holder = nn.Conv2d(w.shape[1], co, 1, 1, 0, bias=True)
        else:
            holder = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True,
                               padding_mode=conv.padding_mode)
        holder.weight.copy_(w)
        holder.bias.copy_(b)
    holder = holder.to(device).requires_grad_(False)
    cache[reshape_1x1_pad_to] = (key, holder)
    return holder


def _pad_out_channels(conv: nn.Conv2d, device, to=32):
    """Same conv with zero output channels appended up to a multiple the MFMA kernel accepts."""
    tensors = [conv.weight, conv.bias]
    key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
    hit = conv.__dict__.get("_dt_padded")
    if hit is not None and hit[0] == key:
        return hit[1]
    co = conv.out_channels
    cop = (co + to - 1) // to * to
    holder = nn.Conv2d(conv.in_channels, cop, conv.kernel_size, conv.stride, conv.padding, bias=True,
                       padding_mode=conv.padding_mode)
    with torch.no_grad():
        holder.weight.zero_()
        holder.bias.zero_()
        holder.weight[:co].copy_(conv.weight)
        holder.bias[:co].copy_(conv.bias)
    holder = holder.to(device).requires_grad_(False)
    conv.__dict__["_dt_padded"] = (key, holder)
    return holder


def stem_im2col(image):
    L = _abi.lib()
    n, c, H, W = image.shape
    if c != 3:
        raise ValueError(f"the matching encoder takes RGB images, got {c} channels")
    img = image.float().contiguous()
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cols = ops.empty_nhwc(n, 152, ho, wo, image.device)
    _abi.check(L.dt_stem_im2col_f32(_abi.ptr(img), _abi.ptr(cols), n, H, W, _abi.current_stream(image.device)),
               "dt_stem_im2col_f32")
    return cols


def stem_conv(image, conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """conv1 + bn1 + relu in one kernel (dt_stem_conv_f32); folded, packed weights cached on the conv."""
    L = _abi.lib()
    n, c, H, W = image.shape
    if c != 3 or tuple(conv.weight.shape) != (64, 3, 7, 7) or conv.stride != (2, 2) or conv.padding != (3, 3):
        raise NotImplementedError("the fused stem is the ResNet conv1: Conv2d(3, 64, 7, stride 2, padding 3)")
    dev = image.device
    folded = _fold_bn(conv, bn, dev)
    ent = folded.__dict__.get("_dt_stem_pack")
    if ent is None:
        packed = torch.empty(int(L.dt_stem_pack_floats()), device=dev, dtype=torch.float32)
        wd = folded.weight.detach().contiguous()
        _abi.check(L.dt_stem_pack_f32(_abi.ptr(wd), _abi.ptr(packed), _abi.current_stream(dev)), "dt_stem_pack_f32")
        ent = (packed, _abi.record_ready(dev))
        folded.__dict__["_dt_stem_pack"] = ent
    else:
        _abi.wait_ready(ent[1], dev)  # packed on another stream that may still be running
    hit = ent[0]
    img = image.float().contiguous()
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = ops.empty_nhwc(n, 64, ho, wo, dev)
    _abi.check(L.dt_stem_conv_f32(_abi.ptr(img), _abi.ptr(hit), _abi.ptr(folded.bias.detach()), _abi.ptr(out), n, H, W,
                                  ops.ACT_RELU, _abi.current_stream(dev)), "dt_stem_conv_f32")
    return out


def maxpool(x, k, stride, pad):
    L = _abi.lib()
    n, c, h, w = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = ops.empty_nhwc(n, c, ho, wo, x.device)
    _abi.check(L.dt_maxpool_f32(_abi.ptr(x), _abi.ptr(out), n, h, w, c, k, stride, pad, _abi.current_stream(x.device)),
               "dt_maxpool_f32")
    return out


def _blur_taps(blur: BlurPool):
    """The 16 filter taps as a host array, read back ONCE per filter version (the kernels take them by value).  Reading
    them on every call cost two device synchronisations per encoder pass -- and cannot be captured into a hipGraph."""
    filt = blur.filt
    key = (filt.data_ptr(), filt._version, str(filt.device))
    hit = blur.__dict__.get("_dt_taps")
    if hit is None or hit[0] != key:
        if not bool((filt == filt[:1]).all()):
            raise NotImplementedError("per-channel blur filters")
        hit = (key, (C.c_float * 16)(*[float(v) for v in filt[0, 0].reshape(-1).tolist()]))
        blur.__dict__["_dt_taps"] = hit
    return hit[1]


def blurpool(x, blur: BlurPool):
    L = _abi.lib()
    n, c, h, w = x.shape
    f16 = _blur_taps(blur)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = ops.empty_nhwc(n, c, ho, wo, x.device)
    _abi.check(L.dt_blurpool4_s2_f32(_abi.ptr(x), _abi.ptr(out), f16, n, h, w, c, _abi.current_stream(x.device)),
               "dt_blurpool4_s2_f32")
    return out


def maxblur(x, blur: BlurPool):
    """MaxPool2d(2, stride 1) + BlurPool(4, stride 2) in one kernel."""
    L = _abi.lib()
    n, c, h, w = x.shape
    f16 = _blur_taps(blur)
    ho, wo = (h - 2) // 2 + 1, (w - 2) // 2 + 1
    out = ops.empty_nhwc(n, c, ho, wo, x.device)
    _abi.check(L.dt_maxblur_f32(_abi.ptr(x), _abi.ptr(out), f16, n, h, w, c, _abi.current_stream(x.device)),
               "dt_maxblur_f32")
    return out


def instance_norm(x, c, eps, act=ops.ACT_NONE, out_nchw=False):
    """x NHWC [n, c_stride, h, w] (channels_last); normalises the first c channels."""
    L = _abi.lib()
    n, cs, h, w = x.shape
    ws = torch.empty(int(L.dt_instnorm_workspace_bytes(n, h * w, c)), dtype=torch.uint8, device=x.device)
    if out_nchw:
        out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    else:
        out = ops.empty_nhwc(n, c, h, w, x.device)
    _abi.check(L.dt_instnorm_f32(_abi.ptr(x), _abi.ptr(out), _abi.ptr(ws), n, h * w, c, cs, float(eps), act, int(out_nchw),
                                 _abi.current_stream(x.device)), "dt_instnorm_f32")
    return out


class ResnetMatchingEncoder(nn.Module):
    """Reference constructor and forward contract (modules/networks.py:141-189); pretrained weights come
    from a loaded state dict (there is no network here to download them)."""

    def __init__(self, num_layers, num_ch_out, pretrained=True, antialiased=True):
        super().__init__()
        if num_layers != 18:
            raise NotImplementedError("only the ResNet-18 matching encoder (the reference's default) is built")
        self.num_ch_enc = np.array([64, 64])
        self.num_ch_out = num_ch_out
        self.antialiased = antialiased
        if antialiased:
            pool = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=1), BlurPool(64, filt_size=4, stride=2))
        else:
            pool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.net = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False),
            nn.BatchNorm2d(64),
            nn.ReLU(inplace=True),
            pool,
            nn.Sequential(ResnetBasicBlock(64), ResnetBasicBlock(64)),
            nn.Conv2d(64, 128, (1, 1)),
            nn.InstanceNorm2d(128),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(128, num_ch_out, (3, 3), padding=1, padding_mode="replicate"),
            nn.InstanceNorm2d(num_ch_out),
        )
        self.eval()

    @torch.no_grad()
    def forward(self, input_image, _impl="mfma", channels_last_output=False):
        if self.training:
            raise NotImplementedError("the HIP matching encoder is inference-only (BatchNorm uses running statistics)")
        net = self.net
        dev = input_image.device
        if _impl == "mfma":
            x = stem_conv(input_image, net[0], net[1])
        else:  # cross-check route: explicit im2col + direct conv
            x = ops.conv2d([(stem_im2col(input_image), False)], _fold_bn(net[0], net[1], dev, reshape_1x1_pad_to=152),
                           act=ops.ACT_RELU, impl=_impl)
        if self.antialiased:
            mp = net[3][0]
            if _impl == "mfma" and (mp.kernel_size, mp.stride, mp.padding) == (2, 1, 0):
                x = maxblur(x, net[3][1])
            else:
                x = maxpool(x, mp.kernel_size, mp.stride, mp.padding)
                x = blurpool(x, net[3][1])
        else:
            x = maxpool(x, net[3].kernel_size, net[3].stride, net[3].padding)
        for blk in net[4]:
            y = ops.conv2d([(x, False)], _fold_bn(blk.conv1, blk.bn1, dev), act=ops.ACT_RELU, impl=_impl)
            x = ops.conv2d([(y, False)], _fold_bn(blk.conv2, blk.bn2, dev), act=ops.ACT_RELU, residual=x, impl=_impl)
        x = ops.conv2d([(x, False)], net[5], act=ops.ACT_NONE, impl=_impl)
        x = instance_norm(x, 128, net[6].eps, act=ops.ACT_LRELU02)
        x = ops.conv2d([(x, False)], _pad_out_channels(net[8], dev), act=ops.ACT_NONE, impl=_impl)
        return instance_norm(x, self.num_ch_out, net[9].eps, out_nchw=not channels_last_output)
