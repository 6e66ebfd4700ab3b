"""ctypes binding of include/doubletake_hip.h.

The product path has no CPU fallback: if the shared library is missing or a call fails this
raises.  (Importing this module does not need a GPU; launching kernels does.)
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

_lib = None


class DoubletakeHipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """struct dt_conv_desc (include/doubletake_hip.h)."""
    _fields_ = [("n", C.c_int), ("h_out", C.c_int), ("w_out", C.c_int), ("c_out", C.c_int), ("nsrc", C.c_int),
                ("c", C.c_int * 3), ("up", C.c_int * 3), ("ksize", C.c_int), ("stride", C.c_int), ("act", C.c_int),
                ("h_in", C.c_int), ("w_in", C.c_int), ("pad_mode", C.c_int), ("transposed", C.c_int)]


class TsdfThresholds(C.Structure):
    """struct dt_tsdf_thresholds (include/doubletake_hip.h)."""
    _fields_ = [(n, C.c_float) for n in ("trunc", "thr_neg", "thr_pos", "max_depth_h", "min_depth", "depth_range")]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float

# name -> (restype, argtypes); mirrors include/doubletake_hip.h declaration by declaration
SIGNATURES = {
    "dt_version": (_I, []),
    "dt_last_error": (C.c_char_p, []),
    "dt_device_count": (_I, []),
    "dt_kernel_launch_count": (_L, []),
    "dt_settings_token": (_L, []),
    "dt_nchw_to_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dt_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dt_cv_params_floats": (_I, [_I, _I]),
    "dt_cv_setup_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "dt_cv_relative_poses_f32": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "dt_cv_warp_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "dt_cv_dot_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_dot_direct_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_dot_stats_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dt_cv_mlp_pack_floats": (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "dt_cv_mlp_hint_f32": (_I, [_P] * 11 + [_I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_mlp_plan_bytes": (_L, [_I, _I, _I, _I]),
    "dt_cv_mlp_set_cu_budget": (_I, [_I]),
    "dt_cv_mlp_plan_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "dt_cv_mlp_hint_planned_f32": (_I, [_P] * 11 + [_I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "dt_cv_mlp_split_pack_halves": (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "dt_cv_mlp_hint_split_f32": (_I, [_P] * 11 + [_I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_mlp_hint_simple_f32": (_I, [_P] * 14 + [_I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_dot_simple_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_lowest_cost_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_cv_overall_mask_u8": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dt_conv_pack_floats": (_L, [_I, _I, _I]),
    "dt_conv_pack_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "dt_conv2d_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "dt_conv_transposed_tiling": (_I, [C.POINTER(ConvDesc)]),
    "dt_conv_set_plan_objective": (_I, [_I]),
    "dt_conv_wino_pack_floats": (_L, [_I, _I]),
    "dt_conv_wino_pack_f32": (_I, [_P, _P, _I, _I, _P]),
    "dt_conv2d_wino_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "dt_conv_wino_split_pack_halves": (_L, [_I, _I]),
    "dt_conv_wino_split_pack_f16": (_I, [_P, _P, _I, _I, _P]),
    "dt_conv2d_wino_split_supported": (_I, [C.POINTER(ConvDesc)]),
    "dt_conv2d_wino_split_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "dt_conv2d_pair_f32": (_I, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "dt_conv2d_simple_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "dt_conv1x1_head_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _P]),
    "dt_head_mlp_pack_floats": (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "dt_head_mlp_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "dt_head_mlp_multi_f32": (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                   C.POINTER(_P), C.POINTER(_L), C.POINTER(_I), _P]),
    "dt_conv2d_wino_heads_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _I, C.POINTER(_P), C.POINTER(_P),
                                      C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_L), C.POINTER(_I), _P]),
    "dt_upsample2x_bilinear_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dt_exp_f32": (_I, [_P, _P, _L, _P]),
    "dt_stem_im2col_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "dt_stem_pack_floats": (_I, []),
    "dt_stem_pack_f32": (_I, [_P, _P, _P]),
    "dt_stem_conv_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dt_maxpool_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dt_blurpool4_s2_f32": (_I, [_P, _P, C.POINTER(_F), _I, _I, _I, _I, _P]),
    "dt_maxblur_f32": (_I, [_P, _P, C.POINTER(_F), _I, _I, _I, _I, _P]),
    "dt_instnorm_workspace_bytes": (_L, [_I, _I, _I]),
    "dt_instnorm_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P]),
    "dt_raster_depth_f32": (_I, [_P, _P, _L, _P, _P, _I, _I, _P, _P, _P]),
    "dt_raster_soup_depth_f32": (_I, [_P, _L, C.POINTER(_F), _F, _P, _P, _I, _I, _P, _P, _P]),
    "dt_hint_from_depth_f32": (_I, [_P, _P, C.POINTER(_F), _F, _I, _I, _I, _P, _P, _F, _I, _I, _P, _P, _P, _P, _I, _P]),
    "dt_tsdf_frame_params_floats": (_I, []),
    "dt_tsdf_frame_setup_f16": (_I, [_P, _P, _I, _I, _F, _F, _P, _P]),
    "dt_tsdf_integrate_f16": (_I, [_P, _P, _P, C.POINTER(_F), _F, _I, _I, _I, _P, _I, _I, _P,
                                   C.POINTER(TsdfThresholds), _P]),
    "dt_tsdf_frames_setup_f16": (_I, [_P, _P, _I, _I, _I, _F, _F, _P, _P]),
    "dt_tsdf_integrate_frames_f16": (_I, [_P, _P, _P, C.POINTER(_F), _F, _I, _I, _I, _P, _I, _I, _I, _P,
                                          C.POINTER(TsdfThresholds), _P]),
    "dt_tsdf_integrate_frames_f32depth_f16": (_I, [_P, _P, _P, C.POINTER(_F), _F, _I, _I, _I, _P, _I, _I, _I, _P,
                                                   C.POINTER(TsdfThresholds), _P]),
    "dt_tsdf_integrate_frames_xslab_f16": (_I, [_P, _P, _P, C.POINTER(_F), _F, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P,
                                                C.POINTER(TsdfThresholds), _P]),
    "dt_tsdf_sample_f16": (_I, [_P, C.POINTER(_F), _F, _I, _I, _I, _P, _P, _L, _I, _P]),
    "dt_sparse_block_voxels": (_I, []),
    "dt_sparse_integrate_f32": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _I, _P, _I, _I, C.POINTER(_F), C.POINTER(_F), _F, _F, _I, _P]),
    "dt_sparse_integrate_frames_f32": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _F, _F, _I, _P]),
    "dt_sparse_sample_f32": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _I, _P, _P, _L, _I, _P]),
    "dt_sparse_mc_count": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P]),
    "dt_sparse_mc_generate": (_I, [_P, _P, _I, _F, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _P, _P, _I, _P]),
    "dt_mc_workspace_bytes": (_L, [_I, _I, _I]),
    "dt_mc_count": (_I, [_P, _P, _I, _I, _I, _F, C.POINTER(_I), C.POINTER(_I), _P, _P, _P]),
    "dt_mc_generate": (_I, [_P, _P, _I, _I, _I, _F, C.POINTER(_I), C.POINTER(_I), _P, _P, _P, _P, _I, _P]),
    "dt_conv_wino4_pack_floats": (_L, [_I, _I]),
    "dt_conv_wino4_pack_f32": (_I, [_P, _P, _I, _I, _P]),
    "dt_conv2d_wino4_blocks": (_L, [C.POINTER(ConvDesc)]),
    "dt_conv2d_wino4_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "dt_program_begin": (_I, [_P]),
    "dt_program_input": (_I, [_P, _L]),
    "dt_program_mark": (_I, []),
    "dt_program_end": (_I, [C.POINTER(_P)]),
    "dt_program_abort": (_I, []),
    "dt_program_launch": (_I, [_P, _I, C.POINTER(_P), _I, _P]),
    "dt_program_info": (_L, [_P, _I]),
    "dt_program_free": (_I, [_P]),
    "dt_mc_raster_depth_f32": (_I, [_P, _P, _I, _I, _I, _F, C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _F, _P, _P, _I, _I,
                                    _P, _P, _P]),
}


# = DT_ABI_VERSION of include/doubletake_hip.h (tests/test_abi.py compares the two)
ABI_VERSION = 105


def lib():
    """Load (building first if the sources changed and hipcc is available) and return the CDLL."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    override = os.environ.get("DOUBLETAKE_HIP_LIB")  # experiment hook: load a differently-built .so
    if override:
        path = override
    elif not _build.is_current():
        try:
            _build.build(verbose=False)
        except Exception as e:  # no hipcc / compile error
            if not os.path.isfile(path):
                raise DoubletakeHipError(
                    f"libdoubletake_hip.so is not built and cannot be built here ({e}). "
                    "Run `python -m doubletake_amd._build` where hipcc is available."
                ) from e
            raise
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first so both sides share one HIP runtime)
    except Exception:
        pass
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.dt_version.restype = _I
    got = L.dt_version()
    if got != ABI_VERSION:  # e.g. a stale variant .so behind DOUBLETAKE_HIP_LIB: shifted arguments, not a clean error
        raise DoubletakeHipError(f"{path}: ABI version {got}, these bindings are written against {ABI_VERSION} "
                                 "(include/doubletake_hip.h: DT_ABI_VERSION) -- rebuild the library")
    ns = _Lib(L)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
        # entry points whose last argument is the stream launch kernels: give them the device guard
        setattr(ns, name, _guarded(fn) if args and args[-1] is _P and name not in _NO_STREAM else fn)
    _lib = ns
    return ns


class _Lib:
    """Namespace of the bound entry points (``lib().dt_conv2d_f32(...)``); ``.cdll`` is the ctypes library."""

    def __init__(self, cdll):
        self.cdll = cdll


class _ForeignStream(C.c_void_p):
    """Stream handle of a GPU that is not HIP's current device (carries the device index for the launch guard)."""
    device_index = None


#: entry points whose trailing void* is a data pointer, not a stream
_NO_STREAM = frozenset({"dt_conv2d_wino_split_supported", "dt_program_free"})


def _guarded(fn):
    def call(*args):
        s = args[-1]
        if type(s) is _ForeignStream:
            with device_guard(s.device_index):
                return fn(*args)
        return fn(*args)

    call.__name__ = fn.__name__
    call.restype, call.argtypes = fn.restype, fn.argtypes
    return call


def check(rc, what=""):
    if rc != 0:
        msg = lib().dt_last_error().decode(errors="replace")
        raise DoubletakeHipError(f"{what}: {msg}" if what else msg)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL) as what a ``c_void_p`` parameter accepts directly: the integer address
    (no ctypes object per argument: ~280 pointer arguments per keyframe)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    """HIP stream handle of torch's current stream on ``device``.  No side effects: when ``device`` is not HIP's current
    device the handle remembers it, and the entry point it is passed to runs under ``device_guard`` (the library
    launches on the *current* device), restoring the caller's device afterwards.
    (Round 5: the raw-handle query ``torch._C._cuda_getCurrentRawStream`` instead of building a ``torch.cuda.Stream`` object per
    launch -- 6 us -> <1 us of host time, ~45 launches per keyframe, and with 4 keyframes in flight the host needs 85 % of a
    step to enqueue it.)"""
    import torch

    idx = None
    if device is not None:
        idx = device.index if type(device) is torch.device else torch.device(device).index
    cur = torch._C._cuda_getDevice()
    if idx is None:
        idx = cur
    handle = torch._C._cuda_getCurrentRawStream(idx)
    if idx != cur:
        s = _ForeignStream(handle)
        s.device_index = idx
        return s
    return C.c_void_p(handle)


class device_guard:
    """``with device_guard(t.device): launch(...)`` -- the library launches on HIP's *current* device (it takes a stream,
    not a device id), so a call with tensors of another GPU switches to that GPU for the duration of the call and
    restores the caller's current device afterwards (the switch used to be permanent).  A no-op -- no HIP
    calls at all -- in the intended one-process-per-GPU deployment, where the tensors' device is the current one."""

    __slots__ = ("idx", "prev")

    def __init__(self, device):
        import torch

        idx = device if isinstance(device, int) else (torch.device(device).index if device is not None else None)
        self.idx = idx
        self.prev = None
        if idx is not None:
            cur = torch.cuda.current_device()
            if cur != idx:
                self.prev = cur

    def __enter__(self):
        if self.prev is not None:
            import torch

            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            import torch

            torch.cuda.set_device(self.prev)
        return False


def record_ready(device):
    """Token kept next to a cached device buffer that a kernel enqueued on the *current* stream is still filling
    (weight packs).  ``wait_ready(token, device)`` on a later cache hit orders another stream behind that kernel."""
    import torch

    st = torch.cuda.current_stream(device)
    ev = torch.cuda.Event()
    ev.record(st)
    return [ev, st.cuda_stream]


def wait_ready(token, device):
    """Cache hit: if the producer kernel of the buffer has not completed and the current stream is not the one it was
    enqueued on, make the current stream wait for it.  Free once the producer is done (the token is cleared)."""
    ev = token[0]
    if ev is None:
        return
    import torch

    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture nothing may look at an outside event; utils/graphs.py warms the caches up (and
        # synchronises) before it captures, so the producer has long finished
        return
    if ev.query():
        token[0] = None
        return
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream != token[1]:
        cur.wait_event(ev)
