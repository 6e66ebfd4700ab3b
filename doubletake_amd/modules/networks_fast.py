"""DoubleTake-small depth decoder (reference modules/networks_fast.py:6-141) on the fused HIP
conv primitive: conv3x3+bias+ELU pairs, nearest x2 upsample and skip concat folded into the
consumer conv's input gather, 1x1 regression heads."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import conv_ops as ops


class ConvBlock(nn.Module):
    def __init__(self, in_ch, out_ch, use_elu=True, use_bn=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, out_ch, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(out_ch, out_ch, kernel_size=3, padding=1)
        if not use_elu:
            raise NotImplementedError("the reference always uses ELU here")
        self.non_lin = nn.ELU(inplace=True)

    def run(self, srcs, impl="mfma"):
        x = ops.conv2d(srcs, self.conv1, act=ops.ACT_ELU, impl=impl)
        return ops.conv2d([(x, False)], self.conv2, act=ops.ACT_ELU, impl=impl)

    @torch.no_grad()
    def forward(self, x):
        return self.run([(ops.as_nhwc(x), False)])


class ConvUpsampleAndConcatBlock(nn.Module):
    def __init__(self, in_ch, out_ch, skip_chns, use_elu=True, use_bn=False):
        super().__init__()
        self.pre_concat_conv = ConvBlock(in_ch, out_ch, use_elu=use_elu, use_bn=use_bn)
        self.post_concat_conv = ConvBlock(out_ch + skip_chns, out_ch, use_elu=use_elu, use_bn=use_bn)

    def run(self, x, cat_feats, impl="mfma"):
        x = self.pre_concat_conv.run([(x, False)], impl=impl)
        # nearest x2 + cat are index transforms inside the next conv's gather
        return self.post_concat_conv.run([(x, True), (cat_feats, False)], impl=impl)

    @torch.no_grad()
    def forward(self, x, cat_feats):
        return self.run(ops.as_nhwc(x), ops.as_nhwc(cat_feats))


class SkipDecoder(nn.Module):
    def __init__(self, input_channels, use_bn=False):
        super().__init__()
        input_channels = list(input_channels)[::-1]
        self.input_channels = input_channels
        self.output_channels = [256, 128, 64, 64]
        self.num_ch_dec = self.output_channels[::-1]
        for bi in range(4):
            setattr(self, f"block{bi + 1}", ConvUpsampleAndConcatBlock(
                in_ch=input_channels[bi], out_ch=self.output_channels[bi], skip_chns=input_channels[bi + 1],
                use_bn=use_bn))

    def _features(self, features, impl="mfma", coarse_heads=None):
        """coarse_heads = (results dict, with_exp) -- SkipDecoderRegression only: the heads of scales 3, 2, 1 read features that
        are final before the last block's 240x320 convolutions start and nothing depends on them, so they are launched as
        extra workgroups of that block's first convolution (ops.conv2d_with_heads) and their results left in the dict."""
        feats = [ops.as_nhwc(f) for f in features]
        out = {}
        x = feats[-1]
        for bi, scale in ((1, 3), (2, 2), (3, 1), (4, 0)):
            blk = getattr(self, f"block{bi}")
            if bi == 4 and coarse_heads is not None and impl == "mfma":
                results, with_exp = coarse_heads
                small = [(s, out[f"feature_s{s}_b1hw"], getattr(self, f"out{4 - s}")) for s in (3, 2, 1)]
                small = [t for t in small if ops.head_mlp_supported(t[1], t[2])
                         and t[1].shape[0] * t[1].shape[2] * t[1].shape[3] <= ops.HEAD_MULTI_MAX_PIXELS]
                x = blk.pre_concat_conv.run([(x, False)], impl=impl)
                post = blk.post_concat_conv
                if len(small) >= 2:
                    y, res = ops.conv2d_with_heads([(x, True), (feats[-1 - bi], False)], post.conv1, ops.ACT_ELU,
                                                   [t[1] for t in small], [t[2] for t in small], with_exp=with_exp)
                    for t, r in zip(small, res):
                        results[t[0]] = r
                else:
                    y = ops.conv2d([(x, True), (feats[-1 - bi], False)], post.conv1, act=ops.ACT_ELU, impl=impl)
                x = ops.conv2d([(y, False)], post.conv2, act=ops.ACT_ELU, impl=impl)
            else:
                x = blk.run(x, feats[-1 - bi], impl=impl)
            out[f"feature_s{scale}_b1hw"] = x
        return out

    @torch.no_grad()
    def forward(self, features):
        return self._features(features)


class SkipDecoderRegression(SkipDecoder):
    def __init__(self, input_channels, use_bn=False):
        super().__init__(input_channels, use_bn=use_bn)
        for oi in range(4):
            setattr(self, f"out{oi + 1}", nn.Sequential(
                nn.Conv2d(self.output_channels[oi], 128, kernel_size=1), nn.ELU(inplace=True),
                nn.Conv2d(128, 128, kernel_size=1), nn.ELU(inplace=True),
                nn.Conv2d(128, 1, kernel_size=1)))

    @torch.no_grad()
    def forward(self, features, _impl="mfma", with_depth=False):
        """Reference contract: {log_depth_pred_s{i}_b1hw, ...}.  with_depth (extension used by
        DepthModelCVHint): the head kernels also write depth_pred_s{i}_b1hw = exp(log depth), saving the four
        exp passes of experiment_modules/doubletake_model.py:410-418."""
        results = {}
        out = self._features(features, impl=_impl, coarse_heads=(results, with_depth) if _impl == "mfma" else None)
        todo = []
        for oi, scale in ((1, 3), (2, 2), (3, 1), (4, 0)):
            head = getattr(self, f"out{oi}")
            feat = out[f"feature_s{scale}_b1hw"]
            todo.append((scale, feat, head, _impl == "mfma" and ops.head_mlp_supported(feat, head)))
        # the coarse heads are independent and each far too small to fill the chip: one launch for all of them (round 5: when
        # _features could place them in the grid of the last block's first convolution, `results` already holds them)
        small = [t for t in todo if t[3] and t[0] not in results
                 and t[1].shape[0] * t[1].shape[2] * t[1].shape[3] <= ops.HEAD_MULTI_MAX_PIXELS]
        if ops.HEAD_MULTI_LAUNCH and len(small) >= 2:
            for t, res in zip(small, ops.head_mlp_multi([t[1] for t in small], [t[2] for t in small], with_exp=with_depth)):
                results[t[0]] = res
        for scale, feat, head, fused in todo:
            if scale in results:
                res = results[scale]
            elif fused:
                # 64- and 128-channel heads (scales 0-2, 99 % of the pixels): one fused kernel
                res = ops.head_mlp(feat, head, with_exp=with_depth)
            else:
                y = ops.conv2d([(feat, False)], head[0], act=ops.ACT_ELU, impl=_impl)
                y = ops.conv2d([(y, False)], head[2], act=ops.ACT_ELU, impl=_impl)
                res = ops.conv1x1_head(y, head[4], with_exp=with_depth)
            if with_depth:
                out[f"log_depth_pred_s{scale}_b1hw"], out[f"depth_pred_s{scale}_b1hw"] = res
            else:
                out[f"log_depth_pred_s{scale}_b1hw"] = res
        return out
