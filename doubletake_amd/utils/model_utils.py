"""Model selection and inference-time loading, reference utils/model_utils.py:10-35 (the training loaders there are out
of scope: SURVEY.md section 2)."""
from __future__ import annotations

import importlib
import io
import pickle
import types
import zipfile

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


# ---- checkpoint reading ---------------------------------------------------------------------------------------------
# Reference checkpoints are Lightning ``.ckpt`` files: ``save_hyperparameters()`` (doubletake_model.py:116,
# sr_depth_model.py:122) pickles the ``doubletake.options`` object into ``hyper_parameters``, next to optimizer states
# and callbacks.  ``torch.load(weights_only=True)`` refuses such a file (unknown global), ``weights_only=False`` would
# import -- i.e. need -- the reference package and run whatever the pickle names.  Only ``state_dict`` matters here, so
# the file is read with an unpickler that resolves tensors / containers normally and turns every other global into an
# inert placeholder class: nothing outside the allow-list below is ever imported or called.
_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
_SAFE_MODULES = {"collections": {"OrderedDict", "defaultdict", "deque"}, "copyreg": {"_reconstructor"},
                 "_codecs": {"encode"}, "argparse": {"Namespace"}}
_TORCH_MODULES = {"torch._utils", "torch._tensor", "torch.nn.parameter", "torch.storage", "torch.serialization"}
_NUMPY_NAMES = {"_reconstruct", "ndarray", "dtype", "scalar", "_frombuffer"}


class _Placeholder:
    """Stand-in for an object of a class that is not loaded (e.g. ``doubletake.options.Options``): accepts any
    construction / state and keeps the state for inspection."""

    def __init__(self, *args, **kwargs):
        self._placeholder_args = (args, kwargs)

    def __setstate__(self, state):
        self.__dict__["_placeholder_state"] = state

    def __call__(self, *args, **kwargs):  # a foreign *function* used as a reduce callable returns a placeholder too
        return _Placeholder(*args, **kwargs)


def _placeholder_class(module, name):
    return type(name, (_Placeholder,), {"__module__": module})


class _StateDictUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "builtins" and name in _SAFE_BUILTINS:
            return super().find_class(module, name)
        if name in _SAFE_MODULES.get(module, ()):
            return super().find_class(module, name)
        if module in _TORCH_MODULES and name.startswith("_rebuild") or (module, name) == ("torch.nn.parameter", "Parameter"):
            return getattr(importlib.import_module(module), name)
        if module == "torch":
            obj = getattr(torch, name, None)
            if isinstance(obj, (type, torch.dtype)):  # storage / tensor classes, torch.Size, torch.device, dtypes
                return obj
        if module.split(".")[0] == "numpy" and name in _NUMPY_NAMES:
            return getattr(importlib.import_module(module), name)
        return _placeholder_class(module, name)


def _restricted_load(f, **kw):
    return _StateDictUnpickler(f, **kw).load()


def _restricted_loads(b, **kw):
    return _StateDictUnpickler(io.BytesIO(b), **kw).load()


# torch.load's legacy (non-zip) reader calls ``pickle_module.load(f)`` directly for the magic number, protocol version,
# sys_info and storage keys, before and after it uses ``pickle_module.Unpickler``: EVERY entry point of the namespace
# therefore goes through the restricted unpickler (a file holding a bare ``pickle.dumps(obj)`` with a hostile
# ``__reduce__`` would otherwise run its callable before "Invalid magic number" is raised).
_state_dict_pickle = types.SimpleNamespace(Unpickler=_StateDictUnpickler, load=_restricted_load, loads=_restricted_loads,
                                           __name__="doubletake_amd.state_dict_pickle")


def read_checkpoint_state_dict(path):
    """``state_dict`` of a checkpoint file written by the reference (Lightning ``.ckpt``) or by ``torch.save`` of a
    plain state dict; tensors on the CPU.  Foreign objects in the file are never instantiated (see above)."""
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            if any(n.endswith("/constants.pkl") or n == "constants.pkl" for n in z.namelist()):
                # torch.load would dispatch such an archive to torch.jit.load (code in the file); never a state dict
                raise RuntimeError(f"{path}: TorchScript archive, not a checkpoint with a state_dict")
    ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_state_dict_pickle)
    state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    if not isinstance(state, dict) or not all(isinstance(v, torch.Tensor) for v in state.values()):
        raise RuntimeError(f"{path}: no tensor state_dict found")
    return state


def load_model_inference(opts, model_class_to_use):
    """utils/model_utils.py:20-35.  The reference first tries Lightning's ``load_from_checkpoint`` and falls back to
    ``model_class(opts)`` + ``load_state_dict(torch.load(path)["state_dict"])``; only the fallback exists here (no
    Lightning).  Keys of modules this package does not own (the timm image encoder ``encoder.*``, the losses) are
    returned as ``model.unused_checkpoint_keys`` instead of raising, and the file is read by
    ``read_checkpoint_state_dict`` (a Lightning checkpoint with pickled ``hyper_parameters`` loads without the reference
    package), so a reference checkpoint loads as it is.
    ``opts.fast_cost_volume`` swaps an MLP volume for its ``to_fast()`` twin exactly as the reference does."""
    model = model_class_to_use(opts)
    path = getattr(opts, "load_weights_from_checkpoint", None)
    if path is not None:
        state = read_checkpoint_state_dict(path)
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
