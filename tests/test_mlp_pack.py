"""Host-side weight packing (doubletake_amd/modules/mlp_pack.py) is a pure permutation of the
reference MLP / head weights: emulate the kernels' contraction order on CPU and compare with the oracle."""
import numpy as np
import pytest

from doubletake_amd.modules import mlp_pack as mp
from doubletake_amd.utils import synthetic as syn
from oracle import cost_volume_ref as cref
from oracle import networks_ref as nref


@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("K", [1, 2, 3, 4, 7, 8, 12])
def test_matching_mlp_pack_is_a_permutation(K, paired):
    """paired = the round-5 layout (metadata of views (1,2), (3,4), ... share a step: 25 instead of 28 metadata steps at K = 7)."""
    cin = syn.mlp_in_channels(K)
    assert cin == mp.Columns(K).total
    p = syn.formula_params(syn.mlp_param_shapes([cin, 128, 128, 1]), 5)
    packed = mp.pack_mlp(*p, K, paired=paired)
    x = syn.hash_normalish((64, cin), 3)
    want = cref.mlp_forward(x, [(p[0], p[1]), (p[2], p[3]), (p[4], p[5])])[:, 0]
    np.testing.assert_allclose(mp.emulate_packed_mlp(packed, x, K, paired=paired), want, atol=2e-6)
    assert packed["w1dyn"].size == mp.dyn_steps_total(K, paired) * 256
    assert mp.dyn_steps_total(7, True) == 81 and mp.dyn_steps_total(7, False) == 84
    # every reference column is fed exactly once
    cols = np.concatenate([mp.dyn_step_columns(K, paired).reshape(-1), mp.pix_step_columns(K).reshape(-1)])
    used = sorted(c for c in cols.tolist() if c >= 0)
    assert used == list(range(cin))
    assert (cols == mp.BIAS).sum() == 1


def test_pack_rejects_wrong_shapes():
    with pytest.raises(ValueError):
        mp.pack_mlp(np.zeros((128, 10), np.float32), np.zeros(128), np.zeros((128, 128)), np.zeros(128), np.zeros((1, 128)), np.zeros(1), 7)
    with pytest.raises(ValueError):
        mp.pack_hint_mlp(np.zeros((12, 4)), np.zeros(12), np.zeros((12, 12)), np.zeros(12), np.zeros((1, 12)), np.zeros(1))
    with pytest.raises(ValueError):
        mp.pack_head_mlp(np.zeros((128, 32, 1, 1)), np.zeros(128), np.zeros((128, 128, 1, 1)), np.zeros(128), np.zeros((1, 128, 1, 1)), np.zeros(1))


@pytest.mark.parametrize("cin", [64, 128])
def test_head_pack_is_a_permutation(cin):
    p = syn.formula_params([(128, cin, 1, 1), (128,), (128, 128, 1, 1), (128,), (1, 128, 1, 1), (1,)], 9)
    pk = mp.pack_head_mlp(*p)
    x = syn.hash_normalish((40, cin), 2)
    wa = pk["wa"].reshape(cin // 2, 2, 32, 4)
    wb = pk["wb"].reshape(64, 2, 32, 4)
    tail = pk["tail"]
    N = x.shape[0]

    def acc_init(off):
        a = np.zeros((N, 128), np.float32)
        for h in range(2):
            for b in range(4):
                for r in range(16):
                    a[:, mp.acc_feature(b, r, h)] = tail[off + h * 64 + b * 16 + r]
        return a

    a1 = acc_init(0)
    for g in range(cin // 8):
        for j in range(4):
            for h in range(2):
                a1 += x[:, 8 * g + 4 * h + j][:, None] * wa[4 * g + j, h].T.reshape(128)[None]
    h1 = nref.elu(a1)
    a2 = acc_init(128)
    for t in range(64):
        for h in range(2):
            a2 += h1[:, mp.acc_feature(t >> 4, t & 15, h)][:, None] * wb[t, h].T.reshape(128)[None]
    h2 = nref.elu(a2)
    s = np.full(N, tail[384], np.float32)
    for h in range(2):
        for b in range(4):
            for r in range(16):
                s += tail[256 + h * 64 + b * 16 + r] * h2[:, mp.acc_feature(b, r, h)]
    xin = x.T.reshape(1, cin, 5, 8)
    y = nref.elu(nref.conv2d(xin, p[0], p[1]))
    y = nref.elu(nref.conv2d(y, p[2], p[3]))
    y = nref.conv2d(y, p[4], p[5])
    np.testing.assert_allclose(s, y.reshape(-1), atol=3e-6)


@pytest.mark.parametrize("K", [1, 2, 7])
def test_split_precision_pack_is_a_permutation(K):
    """Slot tables of the opt-in fp16 hi/lo kernel (csrc/cv_mlp_split.hip): every reference column fed exactly once, and
    the recombined hi + lo fragments reproduce the MLP to fp16-pair precision (weights carry 22 significant bits)."""
    cin = syn.mlp_in_channels(K)
    p = syn.formula_params(syn.mlp_param_shapes([cin, 128, 128, 1]), 5)
    packed = mp.pack_mlp_split(*p, K)
    assert packed["w1dyn"].dtype == np.uint16 and packed["w1dyn"].size == (K + (K + 1) // 2) * 2 * 4 * 64 * 8
    x = syn.hash_normalish((64, cin), 3)
    want = cref.mlp_forward(x, [(p[0], p[1]), (p[2], p[3]), (p[4], p[5])])[:, 0]
    np.testing.assert_allclose(mp.emulate_split_mlp(packed, x, K), want, atol=2e-5)
    cols = np.concatenate([mp.split_dyn_columns(K).reshape(-1), mp.split_pix_columns(K).reshape(-1)])
    assert sorted(c for c in cols.tolist() if c >= 0) == list(range(cin))
    assert (cols == mp.BIAS).sum() == 1
    assert sorted(mp.split_w2_columns().reshape(-1).tolist()) == list(range(128))
