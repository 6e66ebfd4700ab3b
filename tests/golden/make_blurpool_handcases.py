#!/usr/bin/env python3
"""Hand-derived known-answer cases for the anti-aliased ResNet stem's `maxpool`.

The reference builds its matching encoder from `antialiased_cnns.resnet18(pretrained)` (modules/networks.py:158-176;
antialiased-cnns is a pip dependency in environment.yml:29, absent from /root/reference and from this image) and keeps
conv1, bn1, relu, maxpool, layer1.  Everything but `maxpool` is a stock torch.nn layer; `maxpool` is the package's own

    nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=1), BlurPool(64, filt_size=4, stride=2))      (resnet.py, pool_only=True)

with BlurPool (blurpool.py; Zhang, "Making Convolutional Networks Shift-Invariant Again", ICML 2019, section 3 MaxBlurPool):
 B1  filter: a = [1, 3, 3, 1] (filt_size 4), filt = outer(a, a) / sum = outer(a, a) / 64, the same for every channel;
 B2  padding: pad_type 'reflect' (nn.ReflectionPad2d), sizes [int((k-1)/2), ceil((k-1)/2)] = [1, 2] as (left, right) and
     (top, bottom); reflection does not repeat the border sample: index -1 -> 1, n -> n - 2, n + 1 -> n - 3;
 B3  forward: F.conv2d(pad(x), filt, stride=2, groups=C), i.e. out[y][x] = sum_{ky,kx} filt[ky][kx] P[2y+ky][2x+kx];
 M1  the MaxPool2d(2, stride 1) before it has no padding: M[y][x] = max(x[y..y+1][x..x+1]), extent (H-1) x (W-1).

This script evaluates those rules with exact rational arithmetic on small integer images -- no torch, no kernel -- and
writes tests/golden/blurpool_handcases.json.  All expected values are multiples of 1/64 with few bits, so fp32 pipelines
have to reproduce them EXACTLY.  One case is also worked out by hand below and asserted.
"""
import json
import os
from fractions import Fraction as Fr

HERE = os.path.dirname(os.path.abspath(__file__))
A = [1, 3, 3, 1]


def reflect(i, n):  # B2
    if i < 0:
        i = -i
    if i >= n:
        i = 2 * (n - 1) - i
    return i


def max2x2(x):  # M1
    h, w = len(x), len(x[0])
    return [[max(x[i][j], x[i][j + 1], x[i + 1][j], x[i + 1][j + 1]) for j in range(w - 1)] for i in range(h - 1)]


def blurpool(m):  # B1-B3
    h, w = len(m), len(m[0])
    ph, pw = h + 3, w + 3
    P = [[m[reflect(a - 1, h)][reflect(b - 1, w)] for b in range(pw)] for a in range(ph)]
    ho, wo = (ph - 4) // 2 + 1, (pw - 4) // 2 + 1
    return [[sum(Fr(A[ky] * A[kx], 64) * P[2 * y + ky][2 * x + kx] for ky in range(4) for kx in range(4)) for x in range(wo)]
            for y in range(ho)]


def cases():
    cs = []
    # impulse 64 at (1,1) of a 4x4 image.  M (3x3) = 64 on rows/cols {0,1}, 0 on row/col 2.  Padded rows/cols read source
    # indices [1,0,1,2,1,0].  out[0][0] uses [1,0,1,2] in both directions: (1+3+3)(1+3+3) = 49; out[0][1] uses columns
    # [1,2,1,0]: 1+0+3+1 = 5 -> 7*5 = 35; out[1][1] = 25.
    imp = [[0] * 4 for _ in range(4)]
    imp[1][1] = 64
    cs.append(dict(name="impulse_4x4", x=imp, by_hand=[[49, 35], [35, 25]]))
    cs.append(dict(name="ramp_6x6", x=[[(7 * i + 3 * j) % 11 for j in range(6)] for i in range(6)]))
    cs.append(dict(name="odd_7x5", x=[[(5 * i * i + 3 * j + i * j) % 13 - 6 for j in range(5)] for i in range(7)]))
    cs.append(dict(name="constant_8x6", x=[[5] * 6 for _ in range(8)], by_hand=[[5] * 3 for _ in range(4)]))
    return cs


def main():
    out = []
    for c in cases():
        m = max2x2(c["x"])
        mb = blurpool(m)        # the stem: MaxPool2d(2,1) then BlurPool
        b = blurpool(c["x"])    # BlurPool alone (the kernel exists on its own as well)
        if "by_hand" in c:
            assert mb == [[Fr(v) for v in row] for row in c["by_hand"]], (c["name"], mb)
        f = lambda mm: [[float(v) for v in row] for row in mm]
        assert all(float(v) * 64 == int(float(v) * 64) for row in mb + b for v in row)
        out.append(dict(name=c["name"], x=c["x"], maxpool2_s1=m, maxblur=f(mb), blur=f(b)))
        print(c["name"], "->", len(mb), "x", len(mb[0]))
    json.dump(out, open(os.path.join(HERE, "blurpool_handcases.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
