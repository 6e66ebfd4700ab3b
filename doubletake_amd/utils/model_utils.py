"""Model selection and inference-time loading, reference utils/model_utils.py:10-35 (the training loaders there are out
of scope: SURVEY.md section 2)."""
from __future__ import annotations

import torch

from ..experiment_modules.doubletake_model import DepthModel, DepthModelCVHint
from ..modules import cost_volume as _cv


def get_model_class(opts):
    """utils/model_utils.py:10-17."""
    if opts.model_type == "depth_model":
        return DepthModel
    if opts.model_type == "cv_hint_depth_model":
        return DepthModelCVHint
    raise ValueError(f"Unknown model type: {opts.model_type}")


def load_model_inference(opts, model_class_to_use):
    """utils/model_utils.py:20-35.  The reference first tries Lightning's ``load_from_checkpoint`` and falls back to
    ``model_class(opts)`` + ``load_state_dict(torch.load(path)["state_dict"])``; only the fallback exists here (no
    Lightning).  Keys of modules this package does not own (the timm image encoder ``encoder.*``, the losses) are
    returned as ``model.unused_checkpoint_keys`` instead of raising, so a reference checkpoint loads as it is.
    ``opts.fast_cost_volume`` swaps an MLP volume for its ``to_fast()`` twin exactly as the reference does."""
    model = model_class_to_use(opts)
    path = getattr(opts, "load_weights_from_checkpoint", None)
    if path is not None:
        ckpt = torch.load(path, map_location="cpu")
        state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        res = model.load_state_dict(state, strict=False)
        owned = tuple(n + "." for n, _ in model.named_children())
        missing = [k for k in res.missing_keys if k.startswith(owned)]
        if missing:
            raise RuntimeError(f"checkpoint {path} lacks parameters of the hot-path modules: {missing[:8]}...")
        model.unused_checkpoint_keys = list(res.unexpected_keys)
    if getattr(opts, "fast_cost_volume", False) and isinstance(model.cost_volume, _cv.FeatureVolumeManager):
        # (FeatureMeshHintVolumeManager derives from FeatureVolumeManager here; the reference tests both classes)
        model.cost_volume = model.cost_volume.to_fast()
    return model
