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


def view_step_base(k, paired=True):
    """First plane-dependent layer-1 step of source view k (= dt::mlp_view_step_base in csrc/cv_mlp_mfma.hip).
    Round 5: a view has 7 plane-dependent metadata inputs (mask, z', dot, angle, source ray xyz), i.e. 3.5 two-slot K steps.
    Every view runs the same three steps (z'|dot, angle|ray.x, ray.y|ray.z); the mask rides in a fourth step whose other slot
    carries the plane depth (view 0) or the mask of the NEXT view -- the further views are PAIRED (1,2), (3,4), ..., and the
    partner's mask is already known when the step is issued: the kernel has projected the next view by then -- so the second
    view of a pair has no fourth step.  25 instead of 28 metadata steps at K = 7:
    12 MFMAs less per (pixel tile, plane) and 3 KB less LDS.  An unpaired last view keeps four steps (one slot unused)."""
    if k <= 0:
        return 0
    if not paired:  # (library built with -DDT_MLP_PAIR_META=0: four metadata steps for every view)
        return k * STEPS_PER_VIEW
    return STEPS_PER_VIEW + (2 * STEPS_PER_VIEW - 1) * ((k - 1) // 2) + (STEPS_PER_VIEW if (k - 1) % 2 else 0)


def dyn_steps_total(K, paired=True):
    return view_step_base(K, paired)


def dyn_step_columns(K, paired=True):
    """[dyn_steps_total(K), 2] column index fed by half 0 / half 1 at every plane-dependent layer-1 step."""
    c = Columns(K)
    tab = np.full((dyn_steps_total(K, paired), 2), ZERO, dtype=np.int64)
    for k in range(K):
        base = view_step_base(k, paired)
        for s in range(8):
            tab[base + s] = (c.warp + k * FEAT + s, c.warp + k * FEAT + 8 + s)
        pair_first = paired and k >= 1 and (k & 1) == 1 and k + 1 < K
        pair_second = paired and k >= 2 and (k & 1) == 0
        # the same three steps for every view; the seventh value (the mask) rides in a fourth step that the second view of a
        # pair does not have: (mask_0 | plane depth), (mask_k | mask_k+1) for the first of a pair, (mask_k | -) if unpaired
        tab[base + 8] = (c.z + k, c.dot + k)
        tab[base + 9] = (c.ang + k, c.sray + 3 * k + 0)
        tab[base + 10] = (c.sray + 3 * k + 1, c.sray + 3 * k + 2)
        if not pair_second:
            tab[base + 11] = (c.mask + k, c.plane if k == 0 else (c.mask + k + 1 if pair_first else ZERO))
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


def pack_mlp(W1, b1, W2, b2, W3, b3, K, paired=True):
    """numpy in, numpy out: dict(w1dyn, w1pix, w2p, tail) of flat float32 arrays.  paired: the metadata-step layout of the
    library in use (see view_step_base; callers ask dt_cv_mlp_pack_floats)."""
    W1 = np.asarray(W1, dtype=np.float32)
    cin = Columns(K).total
    if W1.shape != (HID, cin):
        raise ValueError(f"matching MLP layer 1 must be [{HID}, {cin}] for {K} source views, got {W1.shape}")
    if np.asarray(W2).shape != (HID, HID) or np.asarray(W3).reshape(-1).shape != (HID,):
        raise ValueError("matching MLP must be [Cin,128,128,1]")
    W_ext = np.concatenate([W1, np.zeros((HID, 1), np.float32), np.asarray(b1, np.float32).reshape(HID, 1)], 1)
    w1dyn = _pack_steps(W_ext, dyn_step_columns(K, paired))
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


def emulate_packed_mlp(packed, x_cols, K, paired=True):
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

    acc1 = run(packed["w1pix"], pix_step_columns(K)) + run(packed["w1dyn"], dyn_step_columns(K, paired))
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


# ---- split-precision (fp16 hi/lo) packing for csrc/cv_mlp_split.hip ----------------------------------------------------------
def split_dyn_columns(K):
    """[K + ceil(K/2), 2, 8] column fed by (lane half kb, slot e) at every plane-dependent K16 step: K feature steps
    (slot = warped channel 8*kb + e of view k), then one metadata step per view PAIR (slots 0..3: view 2m, 4..7: view 2m+1;
    half 0: mask, dot*mask, ray.x, ray.z;  half 1: z', angle, ray.y, plane depth (view 0 only))."""
    c = Columns(K)
    tab = np.full((K + (K + 1) // 2, 2, 8), ZERO, dtype=np.int64)
    for k in range(K):
        for kb in range(2):
            for e in range(8):
                tab[k, kb, e] = c.warp + k * FEAT + 8 * kb + e
        m, e0 = K + (k >> 1), 4 * (k & 1)
        tab[m, 0, e0:e0 + 4] = (c.mask + k, c.dot + k, c.sray + 3 * k + 0, c.sray + 3 * k + 2)
        tab[m, 1, e0:e0 + 4] = (c.z + k, c.ang + k, c.sray + 3 * k + 1, c.plane if k == 0 else ZERO)
    return tab


def split_pix_columns(K):
    """Plane-independent K16 steps: step 0 = current features, then [ray.x, ray.y, ray.z, bias, pd_0, R_0, t_0, pd_1, ...]
    sixteen per step (slot index n -> step 1 + n // 16, half (n % 16) // 8, e = n % 8)."""
    c = Columns(K)
    cols = [c.cray + 0, c.cray + 1, c.cray + 2, BIAS]
    for k in range(K):
        cols += [c.pd + k, c.R + k, c.t + k]
    nsteps = 1 + (len(cols) + 15) // 16
    tab = np.full((nsteps, 2, 8), ZERO, dtype=np.int64)
    for kb in range(2):
        for e in range(8):
            tab[0, kb, e] = c.cur + 8 * kb + e
    for n, col in enumerate(cols):
        tab[1 + n // 16, (n % 16) // 8, n % 8] = col
    return tab


def split_w2_columns():
    """Layer 2: step t = 2*block + q, slot (kb, e) = layer-1 accumulator register 8q + e of that block in lane half kb."""
    tab = np.zeros((8, 2, 8), dtype=np.int64)
    for t in range(8):
        for kb in range(2):
            for e in range(8):
                tab[t, kb, e] = acc_feature(t >> 1, 8 * (t & 1) + e, kb)
    return tab


def _pack_split(W_ext, tab):
    """W_ext [128, Cin + 2] (last two columns: zeros, bias); tab [S,2,8].  Returns uint16 [S, 2 parts, 4 blocks, 64 lanes, 8]:
    the fp16 hi and lo parts of A[row i of block cb][slot (kb, e)], lane = kb*32 + i."""
    cin = W_ext.shape[1] - 2
    cols = np.where(tab == ZERO, cin, np.where(tab == BIAS, cin + 1, tab))
    g = W_ext.T[cols]                                    # [S, kb, e, 128]
    g = g.reshape(*cols.shape, 4, 32).transpose(0, 3, 1, 4, 2)  # [S, cb, kb, i, e]
    g = np.ascontiguousarray(g, dtype=np.float32).reshape(cols.shape[0], 4, 64, 8)
    hi = g.astype(np.float16)
    lo = (g - hi.astype(np.float32)).astype(np.float16)
    return np.ascontiguousarray(np.stack([hi, lo], 1)).view(np.uint16)


def pack_mlp_split(W1, b1, W2, b2, W3, b3, K):
    """dict(w1dyn, w1pix, w2: uint16 arrays of fp16 hi/lo weight fragments; tail: float32 as in pack_mlp)."""
    base = pack_mlp(W1, b1, W2, b2, W3, b3, K)  # validates shapes; the tail (b2, W3, b3 in lane-register order) is shared
    W1 = np.asarray(W1, dtype=np.float32)
    W_ext = np.concatenate([W1, np.zeros((HID, 1), np.float32), np.asarray(b1, np.float32).reshape(HID, 1)], 1)
    W2_ext = np.concatenate([np.asarray(W2, np.float32), np.zeros((HID, 2), np.float32)], 1)
    return dict(w1dyn=_pack_split(W_ext, split_dyn_columns(K)).reshape(-1), w1pix=_pack_split(W_ext, split_pix_columns(K)).reshape(-1),
                w2=_pack_split(W2_ext, split_w2_columns()).reshape(-1), tail=base["tail"])


def emulate_split_mlp(packed, x_cols, K):
    """CPU emulation of the split kernel's slot tables from the PACKED fp16 weights (hi + lo recombined in fp32; inputs kept
    in fp32): proves on the CPU that the tables are a permutation of the reference MLP.  x_cols [N, Cin] -> score [N]."""
    c = Columns(K)
    N = x_cols.shape[0]
    x_ext = np.concatenate([x_cols.astype(np.float32), np.zeros((N, 1), np.float32), np.ones((N, 1), np.float32)], 1)

    def weights(buf, S):
        w = buf.view(np.float16).astype(np.float32).reshape(S, 2, 4, 64, 8)
        return w[:, 0] + w[:, 1]  # [S, cb, lane, e]

    def run(buf, tab):
        cols = np.where(tab == ZERO, c.total, np.where(tab == BIAS, c.total + 1, tab))
        w = weights(buf, tab.shape[0])
        acc = np.zeros((N, HID), np.float32)
        for s in range(tab.shape[0]):
            for kb in range(2):
                for e in range(8):
                    wv = w[s, :, kb * 32:(kb + 1) * 32, e].reshape(HID)  # [cb, i] -> feature 32*cb + i
                    acc += x_ext[:, cols[s, kb, e]][:, None] * wv[None]
        return acc

    acc1 = run(packed["w1pix"], split_pix_columns(K)) + run(packed["w1dyn"], split_dyn_columns(K))
    h1 = np.maximum(acc1, 0.01 * acc1)
    tail = packed["tail"]
    acc2 = np.zeros((N, HID), np.float32)
    for h in range(2):
        for blk in range(4):
            for r in range(16):
                acc2[:, acc_feature(blk, r, h)] = tail[h * 64 + blk * 16 + r]
    w2 = weights(packed["w2"], 8)
    tab2 = split_w2_columns()
    for t in range(8):
        for kb in range(2):
            for e in range(8):
                acc2 += h1[:, tab2[t, kb, e]][:, None] * w2[t, :, kb * 32:(kb + 1) * 32, e].reshape(HID)[None]
    h2 = np.maximum(acc2, 0.01 * acc2)
    s = np.zeros(N, np.float32)
    for h in range(2):
        for blk in range(4):
            for r in range(16):
                s += tail[128 + h * 64 + blk * 16 + r] * h2[:, acc_feature(blk, r, h)]
    return s + tail[256]
