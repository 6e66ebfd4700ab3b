"""CPU reference of ResnetMatchingEncoder.forward (reference modules/networks.py:138-189).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): never imported
by the product path.

The reference module is a composition of torch.nn layers, so this restatement is written with
torch.nn.functional on CPU tensors (fp32), one call per reference layer:

    net.0 conv1 7x7/2 (no bias) -> net.1 BatchNorm2d (eval: running stats) -> net.2 ReLU -> net.3 maxpool
    -> net.4 layer1 = 2 x [conv3x3 - bn - relu - conv3x3 - bn, + identity, relu]
    -> net.5 Conv2d(64,128,1) -> net.6 InstanceNorm2d(128) -> net.7 LeakyReLU(0.2)
    -> net.8 Conv2d(128,C,3,padding=1,padding_mode="replicate") -> net.9 InstanceNorm2d(C)

PARITY STATUS: net.5-net.9 are the reference's own lines (networks.py:181-187) and plain torch.nn
semantics.  net.0-net.4 come from `antialiased_cnns.resnet18` / `torchvision.models.resnet18`
(networks.py:158-176), neither of which is installed in this image; their structure is restated
from memory of those packages (torchvision 0.15 resnet.py; antialiased_cnns 0.3 resnet.py + blurpool.py:
maxpool = Sequential(MaxPool2d(2, stride 1), BlurPool(64, filt_size 4, stride 2, reflect pad (1,2,1,2),
binomial [1,3,3,1] filter)).  The one non-torch layer of that stem, the anti-aliased `maxpool`, is pinned to the package's
published rules by the hand-derived exact known-answer cases of tests/golden/make_blurpool_handcases.py (this restatement and
the HIP kernels both reproduce them bit for bit); the composition itself (which children the reference keeps) follows
networks.py:158-176 and cannot be cross-checked against the package here.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, key):
    v = sd[key]
    return v.detach().float().cpu() if torch.is_tensor(v) else torch.from_numpy(np.asarray(v, dtype=np.float32))


def _bn(x, sd, pre, eps=1e-5):
    return F.batch_norm(x, _t(sd, pre + ".running_mean"), _t(sd, pre + ".running_var"), _t(sd, pre + ".weight"),
                        _t(sd, pre + ".bias"), training=False, eps=eps)


def matching_encoder(image_b3hw, sd, antialiased=True):
    """image: numpy/torch [B,3,H,W]; sd: state dict of the encoder (keys `net.*`).  Returns numpy [B,C,H/4,W/4]."""
    x = torch.as_tensor(np.asarray(image_b3hw), dtype=torch.float32)
    with torch.no_grad():
        x = F.conv2d(x, _t(sd, "net.0.weight"), None, stride=2, padding=3)
        x = F.relu(_bn(x, sd, "net.1"))
        if antialiased:
            x = F.max_pool2d(x, kernel_size=2, stride=1)
            filt = _t(sd, "net.3.1.filt")
            x = F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), filt, stride=2, groups=x.shape[1])
        else:
            x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for blk in ("net.4.0", "net.4.1"):
            y = F.conv2d(x, _t(sd, blk + ".conv1.weight"), None, stride=1, padding=1)
            y = F.relu(_bn(y, sd, blk + ".bn1"))
            y = F.conv2d(y, _t(sd, blk + ".conv2.weight"), None, stride=1, padding=1)
            y = _bn(y, sd, blk + ".bn2")
            x = F.relu(y + x)
        x = F.conv2d(x, _t(sd, "net.5.weight"), _t(sd, "net.5.bias"))
        x = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2)
        x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), _t(sd, "net.8.weight"), _t(sd, "net.8.bias"))
        x = F.instance_norm(x, eps=1e-5)
    return x.numpy()
