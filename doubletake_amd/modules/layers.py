"""BasicBlock with the reference's parameter names, executed by the fused HIP conv primitive.

Reference: modules/layers.py:33-94 -- norm_layer defaults to nn.Identity, which turns conv bias ON
and there is NO BatchNorm; activation LeakyReLU(0.2); the shortcut is the identity, a 1x1 conv
(stride 1, channel change) or a 3x3 stride-2 conv.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import conv_ops as ops


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups, bias=bias,
                     dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=bias)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, groups=1, base_width=64, dilation=1, norm_layer=nn.Identity):
        super().__init__()
        if norm_layer is not nn.Identity:
            raise NotImplementedError("the reference only ever builds BasicBlock with norm_layer=nn.Identity")
        if groups != 1 or base_width != 64 or dilation > 1:
            raise ValueError("BasicBlock only supports groups=1, base_width=64, dilation=1")
        self.conv1 = conv3x3(inplanes, planes, stride, bias=True)
        self.bn1 = nn.Identity()
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.conv2 = conv3x3(planes, planes, bias=True)
        self.bn2 = nn.Identity()
        if inplanes == planes and stride == 1:
            self.downsample = None
        else:
            conv = conv1x1 if stride == 1 else conv3x3
            self.downsample = nn.Sequential(conv(inplanes, planes, bias=True, stride=stride), nn.Identity())
        self.stride = stride

    def run(self, srcs, impl="mfma"):
        """srcs: list of (NHWC tensor, nearest_up) forming the (virtually concatenated) block input."""
        if self.downsample is not None and impl == "mfma":
            # conv1 and the shortcut conv read the same input and nothing depends on the shortcut until conv2: one launch
            t, identity = ops.conv2d_pair(srcs, self.conv1, ops.ACT_LRELU02, self.downsample[0], ops.ACT_NONE)
            return ops.conv2d([(t, False)], self.conv2, act=ops.ACT_LRELU02, residual=identity, impl=impl)
        t = ops.conv2d(srcs, self.conv1, act=ops.ACT_LRELU02, impl=impl)
        if self.downsample is not None:
            identity = ops.conv2d(srcs, self.downsample[0], act=ops.ACT_NONE, impl=impl)
        else:
            if len(srcs) != 1 or srcs[0][1]:
                raise ValueError("identity shortcut needs a single, non-upsampled source")
            identity = srcs[0][0]
        return ops.conv2d([(t, False)], self.conv2, act=ops.ACT_LRELU02, residual=identity, impl=impl)

    @torch.no_grad()
    def forward(self, x):
        return self.run([(ops.as_nhwc(x), False)])


class TensorFormatter(nn.Module):
    """modules/layers.py:97-134 (pure reshapes; kept for API parity)."""

    def __init__(self):
        super().__init__()
        self.batch_size = None
        self.depth_chns = None

    def _expand_batch_with_channels(self, x):
        if x.dim() != 5:
            raise ValueError("TensorFormatter expects tensors with 5 dimensions, not {}!".format(len(x.shape)))
        self.batch_size, self.depth_chns, chns, height, width = x.shape
        return x.view(self.batch_size * self.depth_chns, chns, height, width)

    def _reduce_batch_to_channels(self, x):
        if self.batch_size is None or self.depth_chns is None:
            raise ValueError("Cannot call _reduce_batch_to_channels without first calling _expand_batch_with_channels!")
        _, chns, height, width = x.shape
        return x.view(self.batch_size, self.depth_chns, chns, height, width)

    def forward(self, x, apply_func):
        return self._reduce_batch_to_channels(apply_func(self._expand_batch_with_channels(x)))
