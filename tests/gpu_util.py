"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch

from doubletake_amd.utils import synthetic as syn


def dev():
    return torch.device("cuda:0")


def to_dev(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev()) for k, v in d.items()}


def volume_call_args(inp_t):
    return dict(
        cur_feats=inp_t["cur_feats"], src_feats=inp_t["src_feats"], src_extrinsics=inp_t["src_extrinsics"],
        src_poses=inp_t["src_poses"], src_Ks=inp_t["src_Ks"], cur_invK=inp_t["cur_invK"],
        min_depth=inp_t["min_depth"], max_depth=inp_t["max_depth"],
    )


def hint_dict(inp_t):
    return {n: inp_t[n] for n in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")}


def load_formula_mlp(mlp_module, channels, seed):
    """Set nn.Linear params of an MLP container to the golden script's formula weights."""
    params = syn.formula_params(syn.mlp_param_shapes(channels), seed)
    with torch.no_grad():
        for p, a in zip(mlp_module.parameters(), params):
            p.copy_(torch.from_numpy(a))
    return [(params[i], params[i + 1]) for i in range(0, len(params), 2)]


def set_formula_weights(module, seed, scale_mult=1.0):
    """Same rule as tests/golden/make_golden.py:set_formula_weights, applied to our modules."""
    shapes = [tuple(p.shape) for _, p in module.named_parameters()]
    arrs = syn.formula_params(shapes, seed, scale_mult)
    with torch.no_grad():
        for (_, p), a in zip(module.named_parameters(), arrs):
            p.copy_(torch.from_numpy(a).to(p.device))
    return arrs
