"""MLP container with the reference's parameter names (modules/networks.py:120-135).

The CVEncoder / DepthDecoderPP shims live in conv_networks.py and are re-exported here.
"""
from __future__ import annotations

import torch.nn as nn


class MLP(nn.Module):
    """nn.Linear + LeakyReLU(0.01) stack; keys ``net.{0,2,4}.{weight,bias}``.

    In the cost-volume managers this module only *owns* the weights (state-dict
    compatibility); the arithmetic runs inside the fused HIP kernel.
    """

    def __init__(self, channel_list, disable_final_activation=False):
        super().__init__()
        layers = []
        for i in range(len(channel_list) - 1):
            layers.append(nn.Linear(channel_list[i], channel_list[i + 1]))
            layers.append(nn.LeakyReLU(inplace=True))
        if disable_final_activation:
            layers = layers[:-1]
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)
