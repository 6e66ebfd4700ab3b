"""Host-side re-packing of the matching-MLP weights into the K order the fused MFMA kernel
consumes (csrc/cv_mlp_mfma.hip).  Pure index shuffling, done once per weight update.

Reference column order of the MLP input (modules/mesh_hint_volume.py:353-370), K source views,
16 feature channels:
    [0,16K)            warped source features, k-major
    [16K,16K+16)       current features
    +K                 mask_k
    +K                 z'_k
    +1                 plane depth
    +K                 dot_k * mask_k
    +K                 ray angle_k
    +3                 current ray
    +3K                source rays (k, xyz)
    +K, +K, +K         pose distance, R measure, t measure
"""
from __future__ import annotations

import numpy as np

FEAT = 16
HID = 128
STEPS_PER_VIEW = 12
PIX_FIXED = 10
W2_STEPS = 64
TAIL_FLOATS = 260
BIAS = -2  # pseudo column: the layer-1 bias (fed by a constant-1 input)
ZERO = -1  # unused slot


class Columns:
    def __init__(self, K):
        self.K = K
        o = 0
        self.warp = o; o += FEAT * K
        self.cur = o; o += FEAT
        self.mask = o; o += K
        self.z = o; o += K
        self.plane = o; o += 1
        self.dot = o; o += K
        self.ang = o; o += K
        self.cray = o; o += 3
        self.sray = o; o += 3 * K
        self.pd = o; o += K
        self.R = o; o += K
        self.t = o; o += K
        self.total = o


def dyn_step_columns(K):
    """[K*12, 2] column index fed by half 0 / half 1 at every plane-dependent layer-1 step."""
    c = Columns(K)
    tab = np.full((K * STEPS_PER_VIEW, 2), ZERO, dtype=np.int64)
    for k in range(K):
        base = k * STEPS_PER_VIEW
        for s in range(8):
            tab[base + s] = (c.warp + k * FEAT + s, c.warp + k * FEAT + 8 + s)
        tab[base + 8] = (c.mask + k, c.z + k)
        tab[base + 9] = (c.dot + k, c.ang + k)
        tab[base + 10] = (c.sray + 3 * k + 0, c.sray + 3 * k + 1)
        tab[base + 11] = (c.sray + 3 * k + 2, c.plane if k == 0 else ZERO)
    return tab


def pix_step_columns(K):
    """[10 + 2K, 2] columns of the plane-independent steps (contracted once per pixel)."""
    c = Columns(K)
    tab = np.full((PIX_FIXED + 2 * K, 2), ZERO, dtype=np.int64)
    for s in range(8):
        tab[s] = (c.cur + s, c.cur + 8 + s)
    tab[8] = (c.cray + 0, c.cray + 1)
    tab[9] = (c.cray + 2, BIAS)
    for k in range(K):
        tab[PIX_FIXED + 2 * k] = (c.pd + k, c.R + k)
        tab[PIX_FIXED + 2 * k + 1] = (c.t + k, ZERO)
    return tab


def acc_feature(block, r, half):
    """Feature index held by accumulator register r of 32-feature block `block` in lane-half
    `half` (v_mfma_f32_32x32x2_f32 C/D layout: row = (r&3) + 8*(r>>2) + 4*half)."""
    return block * 32 + (r & 3) + 8 * (r >> 2) + 4 * half


def _pack_steps(W_ext, tab):
    """W_ext: [128, Cin + 2] with column Cin = zeros (ZERO) and Cin+1 = bias (BIAS).
    Returns float32 [steps, 2, 32, 4] with [s,h,i,fb] = W_ext[fb*32+i, tab[s,h]]."""
    cin = W_ext.shape[1] - 2
    cols = np.where(tab == ZERO, cin, np.where(tab == BIAS, cin + 1, tab))  # [S,2]
    # gather -> [S,2,128] then split 128 -> (fb, i) -> [S,2,i,fb]
    g = W_ext.T[cols]  # [S,2,128]
    g = g.reshape(cols.shape[0], 2, 4, 32).transpose(0, 1, 3, 2)
    return np.ascontiguousarray(g, dtype=np.float32)


def pack_mlp(W1, b1, W2, b2, W3, b3, K):
    """numpy in, numpy out: dict(w1dyn, w1pix, w2p, tail) of flat float32 arrays."""
    W1 = np.asarray(W1, dtype=np.float32)
    cin = Columns(K).total
    if W1.shape != (HID, cin):
        raise ValueError(f"matching MLP layer 1 must be [{HID}, {cin}] for {K} source views, got {W1.shape}")
    if np.asarray(W2).shape != (HID, HID) or np.asarray(W3).reshape(-1).shape != (HID,):
        raise ValueError("matching MLP must be [Cin,128,128,1]")
    W_ext = np.concatenate([W1, np.zeros((HID, 1), np.float32), np.asarray(b1, np.float32).reshape(HID, 1)], 1)
    w1dyn = _pack_steps(W_ext, dyn_step_columns(K))
    w1pix = _pack_steps(W_ext, pix_step_columns(K))
    # layer 2: step t = block*16 + r, half h feeds input feature acc_feature(block, r, h)
    W2 = np.asarray(W2, dtype=np.float32)
    tab2 = np.zeros((W2_STEPS, 2), dtype=np.int64)
    for t in range(W2_STEPS):
        for h in range(2):
            tab2[t, h] = acc_feature(t >> 4, t & 15, h)
    W2_ext = np.concatenate([W2, np.zeros((HID, 2), np.float32)], 1)
    w2p = _pack_steps(W2_ext, tab2)
    tail = np.zeros(TAIL_FLOATS, dtype=np.float32)
    b2 = np.asarray(b2, np.float32).reshape(-1)
    W3 = np.asarray(W3, np.float32).reshape(-1)
    for h in range(2):
        for blk in range(4):
            for r in range(16):
                f = acc_feature(blk, r, h)
                tail[h * 64 + blk * 16 + r] = b2[f]
                tail[128 + h * 64 + blk * 16 + r] = W3[f]
    tail[256] = np.asarray(b3, np.float32).reshape(-1)[0]
    return dict(w1dyn=w1dyn.reshape(-1), w1pix=w1pix.reshape(-1), w2p=w2p.reshape(-1), tail=tail)


def pack_hint_mlp(V1, c1, V2, c2, V3, c3):
    """217 floats: V1[12x3], c1[12], V2[12x12], c2[12], V3[12], c3."""
    parts = [np.asarray(a, np.float32).reshape(-1) for a in (V1, c1, V2, c2, V3, c3)]
    out = np.concatenate(parts)
    if out.size != 217:
        raise ValueError(f"hint MLP must be [3,12,12,1]; got {out.size} parameters")
    return out


def emulate_packed_mlp(packed, x_cols, K):
    """CPU emulation of the kernel's contraction order from the PACKED weights (used by the
    not-gpu tests to prove the packing tables are a permutation of the reference MLP):
    x_cols [N, Cin] -> matching score [N]."""
    c = Columns(K)
    N = x_cols.shape[0]
    x_ext = np.concatenate([x_cols.astype(np.float32), np.zeros((N, 1), np.float32), np.ones((N, 1), np.float32)], 1)

    def run(wp, tab):
        cols = np.where(tab == ZERO, c.total, np.where(tab == BIAS, c.total + 1, tab))
        wp = wp.reshape(tab.shape[0], 2, 32, 4)
        acc = np.zeros((N, HID), np.float32)
        for s in range(tab.shape[0]):
            for h in range(2):
                w = wp[s, h].T.reshape(HID)  # [fb,i] -> feature fb*32+i
                acc += x_ext[:, cols[s, h]][:, None] * w[None]
        return acc

    acc1 = run(packed["w1pix"], pix_step_columns(K)) + run(packed["w1dyn"], dyn_step_columns(K))
    h1 = np.maximum(acc1, 0.01 * acc1)
    w2 = packed["w2p"].reshape(W2_STEPS, 2, 32, 4)
    tail = packed["tail"]
    acc2 = np.zeros((N, HID), np.float32)
    for h in range(2):
        for blk in range(4):
            for r in range(16):
                acc2[:, acc_feature(blk, r, h)] = tail[h * 64 + blk * 16 + r]
    for t in range(W2_STEPS):
        for h in range(2):
            fin = acc_feature(t >> 4, t & 15, h)
            acc2 += h1[:, fin][:, None] * w2[t, h].T.reshape(HID)[None]
    h2 = np.maximum(acc2, 0.01 * acc2)
    s = np.zeros(N, np.float32)
    for h in range(2):
        for blk in range(4):
            for r in range(16):
                s += tail[128 + h * 64 + blk * 16 + r] * h2[:, acc_feature(blk, r, h)]
    return s + tail[256]


HEAD_TAIL_FLOATS = 388


def pack_head_mlp(Wa, ba, Wb, bb, Wc, bc):
    """Regression head Cin -> 128 -> 128 -> 1 (1x1 convs of SkipDecoderRegression.out*) for
    csrc/head_mlp.hip.  Layer A step 4g+j feeds input channels (8g+j | 8g+4+j) from the two lane
    halves (the halves of the float4 each lane loads); layer B uses the accumulator order."""
    Wa = np.asarray(Wa, np.float32).reshape(HID, -1)
    cin = Wa.shape[1]
    if cin not in (64, 128, 256):
        raise ValueError("fused head supports 64, 128 or 256 input channels")
    Wb = np.asarray(Wb, np.float32).reshape(HID, HID)
    tabA = np.zeros((cin // 2, 2), dtype=np.int64)
    for g in range(cin // 8):
        for j in range(4):
            tabA[4 * g + j] = (8 * g + j, 8 * g + 4 + j)
    wa = _pack_steps(np.concatenate([Wa, np.zeros((HID, 2), np.float32)], 1), tabA)
    tabB = np.zeros((W2_STEPS, 2), dtype=np.int64)
    for t in range(W2_STEPS):
        for h in range(2):
            tabB[t, h] = acc_feature(t >> 4, t & 15, h)
    wb = _pack_steps(np.concatenate([Wb, np.zeros((HID, 2), np.float32)], 1), tabB)
    tail = np.zeros(HEAD_TAIL_FLOATS, dtype=np.float32)
    ba = np.asarray(ba, np.float32).reshape(-1)
    bb = np.asarray(bb, np.float32).reshape(-1)
    wc = np.asarray(Wc, np.float32).reshape(-1)
    for h in range(2):
        for blk in range(4):
            for r in range(16):
                f = acc_feature(blk, r, h)
                tail[h * 64 + blk * 16 + r] = ba[f]
                tail[128 + h * 64 + blk * 16 + r] = bb[f]
                tail[256 + h * 64 + blk * 16 + r] = wc[f]
    tail[384] = np.asarray(bc, np.float32).reshape(-1)[0]
    return dict(wa=wa.reshape(-1), wb=wb.reshape(-1), tail=tail)
