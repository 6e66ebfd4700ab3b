"""View helpers of the volume managers (reference utils/generic_utils.py:111-137) and the host-device
move helper of the drivers (:138-145).  Pure metadata operations: no kernel runs here -- the fused
volume kernels index [b, K, ...] tensors directly, these exist so that code written against the
reference's helpers (e.g. a maintainer's own volume manager subclass) keeps working unchanged."""
from __future__ import annotations

import torch


def tensor_B_to_bM(tensor_BS: torch.Tensor, batch_size: int, num_views: int) -> torch.Tensor:
    """[b*M, ...] -> [b, M, ...] (reference :111-120).  A view: raises if the input is not viewable."""
    return tensor_BS.view(batch_size, num_views, *tensor_BS.shape[1:])


def tensor_bM_to_B(tensor_bMS: torch.Tensor) -> torch.Tensor:
    """[b, M, ...] -> [b*M, ...] (reference :123-131)."""
    b, m = tensor_bMS.shape[:2]
    return tensor_bMS.view(b * m, *tensor_bMS.shape[2:])


def combine_dims(x: torch.Tensor, dim_begin: int, dim_end: int) -> torch.Tensor:
    """Fold dimensions dim_begin..dim_end-1 into one (reference :134-137)."""
    return x.view(*x.shape[:dim_begin], -1, *x.shape[dim_end:])


def to_gpu(input_dict, key_ignores=()):
    """Move every tensor entry of a batch dict to the current GPU, as float (reference :138-145)."""
    for k, v in input_dict.items():
        if k not in key_ignores and torch.is_tensor(v):
            input_dict[k] = v.cuda().float()
    return input_dict
