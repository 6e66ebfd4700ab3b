"""Host-side plumbing for the NHWC fp32-MFMA conv primitive (csrc/conv.hip).

Activations travel between our modules as torch tensors of logical shape [n,C,h,w] in
``torch.channels_last`` memory format, i.e. physically NHWC -- still valid inputs for any
torch op, so every module stays a drop-in at its own boundary.
"""
from __future__ import annotations

import ctypes as C
import os as _os

import torch
import torch.nn as nn

from .. import _abi

ACT_NONE, ACT_LRELU02, ACT_ELU, ACT_RELU = 0, 1, 2, 3


#: optional accounting of the work launched through this module (bench.py ``roofline_conv``): set to a dict with the
#: keys "flops" / "calls" and every op adds its direct-convolution-equivalent FLOPs (2 * outputs * K) to it
ACCOUNT = None


def _account(flops):
    if ACCOUNT is not None:
        ACCOUNT["flops"] += float(flops)
        ACCOUNT["calls"] += 1


def _require_gpu(t, name="input"):
    if not t.is_cuda:
        raise _abi.DoubletakeHipError(
            f"{name} is on {t.device}; doubletake_amd convs only run on a ROCm GPU (no CPU fallback)")


def empty_nhwc(n, c, h, w, device):
    return torch.empty((n, c, h, w), device=device, dtype=torch.float32, memory_format=torch.channels_last)


def _is_nhwc(x):
    n, c, h, w = x.shape
    return x.dtype == torch.float32 and x.stride() == (h * w * c, 1, w * c, c)


def as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Return x as a dense fp32 NHWC (channels_last) tensor, converting with our own kernel if needed."""
    _require_gpu(x)
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D tensor, got {tuple(x.shape)}")
    if _is_nhwc(x):
        return x
    n, c, h, w = x.shape
    src = x.float().contiguous()
    out = empty_nhwc(n, c, h, w, x.device)
    L = _abi.lib()
    _abi.check(L.dt_nchw_to_nhwc_f32(_abi.ptr(src), _abi.ptr(out), n, c, h, w, _abi.current_stream(x.device)),
               "dt_nchw_to_nhwc_f32")
    return out


def packed_weight(conv: nn.Conv2d, device, transposed=False):
    """Device buffer with conv.weight re-packed for the MFMA kernel; cached per weight version.
    transposed: pack W.transpose(2, 3) (for launches with dt_conv_desc.transposed = 1)."""
    w = conv.weight
    key = (w.data_ptr(), w._version, device)
    slot = "_dt_pack_t" if transposed else "_dt_pack"
    hit = getattr(conv, slot, None)
    if hit is not None and hit[0] == key:
        _abi.wait_ready(hit[2], device)  # packed on another stream that may still be running
        return hit[1]
    co, ci, k, k2 = w.shape
    if (k != k2 or conv.groups != 1 or conv.dilation != (1, 1) or conv.padding != (k // 2, k // 2)
            or conv.padding_mode not in ("zeros", "replicate")):
        raise NotImplementedError(f"unsupported conv configuration {conv}")
    L = _abi.lib()
    wd = w.detach().to(device=device, dtype=torch.float32)
    wd = (wd.transpose(2, 3) if transposed else wd).contiguous()
    packed = torch.empty(int(L.dt_conv_pack_floats(co, ci, k)), device=device, dtype=torch.float32)
    _abi.check(L.dt_conv_pack_f32(_abi.ptr(wd), _abi.ptr(packed), co, ci, k, _abi.current_stream(device)),
               "dt_conv_pack_f32")
    setattr(conv, slot, (key, packed, _abi.record_ready(device)))
    return packed


#: tile small maps in the transposed frame when that needs fewer workgroups (DT_CONV_TRANSPOSE=0 disables; A/B switch)
TRANSPOSED_TILING = _os.environ.get("DT_CONV_TRANSPOSE", "1") != "0"


_TRANSPOSE_MEMO = {}


def _want_transposed(L, d) -> bool:
    """dt_conv_transposed_tiling(d), memoised on the launch shape (one ctypes call less per conv launch)."""
    if not TRANSPOSED_TILING:
        return False
    key = (d.n, d.h_out, d.w_out, d.c_out, d.nsrc, d.c[0], d.c[1], d.c[2], d.ksize, d.stride)
    hit = _TRANSPOSE_MEMO.get(key)
    if hit is None:
        hit = _TRANSPOSE_MEMO[key] = bool(L.dt_conv_transposed_tiling(C.byref(d)))
    return hit


def packed_weight_wino(conv: nn.Conv2d, device):
    """Winograd-domain weights (G g G^T) of a 3x3 conv, packed for conv_wino_kernel; cached per weight version."""
    w = conv.weight
    key = (w.data_ptr(), w._version, device)
    hit = getattr(conv, "_dt_pack_wino", None)
    if hit is not None and hit[0] == key:
        _abi.wait_ready(hit[2], device)
        return hit[1]
    co, ci, k, k2 = w.shape
    if (k, k2) != (3, 3) or conv.stride != (1, 1) or conv.groups != 1 or conv.dilation != (1, 1) or conv.padding != (1, 1):
        raise NotImplementedError(f"Winograd path needs a 3x3 stride-1 pad-1 conv, got {conv}")
    L = _abi.lib()
    wd = w.detach().to(device=device, dtype=torch.float32).contiguous()
    packed = torch.empty(int(L.dt_conv_wino_pack_floats(co, ci)), device=device, dtype=torch.float32)
    _abi.check(L.dt_conv_wino_pack_f32(_abi.ptr(wd), _abi.ptr(packed), co, ci, _abi.current_stream(device)),
               "dt_conv_wino_pack_f32")
    conv._dt_pack_wino = (key, packed, _abi.record_ready(device))
    return packed


def packed_weight_wino_split(conv: nn.Conv2d, device):
    """Winograd-domain weights as fp16 hi/lo fragments for the opt-in split-precision kernel; cached per weight version."""
    w = conv.weight
    key = (w.data_ptr(), w._version, device)
    hit = getattr(conv, "_dt_pack_wino_split", None)
    if hit is not None and hit[0] == key:
        _abi.wait_ready(hit[2], device)
        return hit[1]
    co, ci, k, k2 = w.shape
    if (k, k2) != (3, 3) or conv.stride != (1, 1) or conv.groups != 1 or conv.dilation != (1, 1) or conv.padding != (1, 1):
        raise NotImplementedError(f"Winograd path needs a 3x3 stride-1 pad-1 conv, got {conv}")
    L = _abi.lib()
    wd = w.detach().to(device=device, dtype=torch.float32).contiguous()
    packed = torch.empty(int(L.dt_conv_wino_split_pack_halves(co, ci)), device=device, dtype=torch.int16)
    _abi.check(L.dt_conv_wino_split_pack_f16(_abi.ptr(wd), _abi.ptr(packed), co, ci, _abi.current_stream(device)),
               "dt_conv_wino_split_pack_f16")
    conv._dt_pack_wino_split = (key, packed, _abi.record_ready(device))
    return packed


def packed_weight_wino4(conv: nn.Conv2d, device):
    """F(4x4, 3x3) Winograd-domain weights (G g G^T, 36 positions) packed for conv_wino4_kernel; cached per weight version."""
    w = conv.weight
    key = (w.data_ptr(), w._version, device)
    hit = getattr(conv, "_dt_pack_wino4", None)
    if hit is not None and hit[0] == key:
        _abi.wait_ready(hit[2], device)
        return hit[1]
    co, ci, k, k2 = w.shape
    if (k, k2) != (3, 3) or conv.stride != (1, 1) or conv.groups != 1 or conv.dilation != (1, 1) or conv.padding != (1, 1):
        raise NotImplementedError(f"Winograd path needs a 3x3 stride-1 pad-1 conv, got {conv}")
    L = _abi.lib()
    wd = w.detach().to(device=device, dtype=torch.float32).contiguous()
    packed = torch.empty(int(L.dt_conv_wino4_pack_floats(co, ci)), device=device, dtype=torch.float32)
    _abi.check(L.dt_conv_wino4_pack_f32(_abi.ptr(wd), _abi.ptr(packed), co, ci, _abi.current_stream(device)),
               "dt_conv_wino4_pack_f32")
    conv._dt_pack_wino4 = (key, packed, _abi.record_ready(device))
    return packed


#: OPT-IN: F(4x4, 3x3) instead of F(2x2, 3x3) for 3x3 stride-1 layers whose launch has at least this many 16x16-pixel x
#: 32-channel workgroups (csrc/conv_wino4.hip: 2.25 instead of 4 multiplies per output pixel).  0 (default) = never: the kernel
#: is correct and tested, but on gfx950 it measured 0.7-0.85x the speed of the F(2x2) kernel on every layer shape of the conv
#: stacks at batch 1 and 8 (profiles/r6i_wino4_ab2.txt) -- neither kernel is bound by the matrix pipe (the fp32 MFMAs and the
#: transform's vector instructions do not overlap, and F(4x4) has 2.3x the vector work per MFMA), see DESIGN.md 4.8.
#: DT_CONV_WINO4_MIN_BLOCKS overrides.
WINO4_MIN_BLOCKS = int(_os.environ.get("DT_CONV_WINO4_MIN_BLOCKS", "0"))


#: arithmetic of the Winograd-domain products of the 3x3 stride-1 layers: "fp32" (default, exact fp32 MFMA) or the opt-in
#: "split16" (fp16 hi/lo operand pairs on the fp16 matrix pipe, fp32 accumulation; csrc/conv_wino_split.hip).  Layers the
#: split kernel does not take (fewer than SPLIT_MIN_BLOCKS workgroups, a source that is not a multiple of 16 channels,
#: stride 2, 1x1) stay on the fp32 kernels.  DT_CONV_PRECISION sets the initial value.
CONV_PRECISION = _os.environ.get("DT_CONV_PRECISION", "fp32")
SPLIT_MIN_BLOCKS = int(_os.environ.get("DT_CONV_SPLIT_MIN_BLOCKS", "128"))


#: use the Winograd kernel for 3x3 stride-1 layers with at least this many 8x16-pixel x 32-channel workgroups
#: (set by measurement, see DESIGN.md section 4.2: 96 = down to the 30x40 level, whose 96 blocks run as 2 workgroups each;
#: 48 = the 15x20 level too measured slower); DT_CONV_WINO_MIN_BLOCKS overrides, 0 disables
WINO_MIN_BLOCKS = int(_os.environ.get("DT_CONV_WINO_MIN_BLOCKS", "96"))


#: Plan objective of the conv launchers (include/doubletake_hip.h: dt_conv_set_plan_objective).  PLAN_LATENCY: every launch as
#: short as possible on an otherwise idle chip (one keyframe at a time: the incremental mode).  PLAN_THROUGHPUT: several
#: independent keyframes in flight on HIP streams -- no in-workgroup K split of the Winograd layers, 4 instead of 8 waves in the
#: direct K-split kernels, no tail split (bits 1 + 2 + 8), and the Winograd kernel down to the 15x20 level (fewer MFMAs; slower
#: as a lone launch) -- so that another frame's workgroups fit beside them.  bench.py with 4 keyframes in flight: 745 -> 774-779
#: frames/s, single stream 1.69 -> 1.80-1.85 ms (profiles/r5b_conv_env_probe.txt, r5c_conv_obj_probe.txt: every bit and
#: combination measured; switching the cross-workgroup K split off as well, bit 4, loses what the others gain).  Same products
#: either way (the fp32 summation order of a K split can differ); process-wide.
PLAN_LATENCY = 0
PLAN_THROUGHPUT = 11
_WINO_MIN_BLOCKS_BY_PLAN = {False: 96, True: 48}  # (throughput plan: True)


_PLAN_MASK = int(_os.environ.get("DT_CONV_OBJ", "0") or 0)  # (the library presets itself from the same variable)


def current_plan_objective():
    """The objective mask in force (what the last set_plan_objective returned; DT_CONV_OBJ before the first call)."""
    return _PLAN_MASK


def set_plan_objective(mask):
    """Select the plan objective (PLAN_LATENCY / PLAN_THROUGHPUT or a bit mask, see the header); returns the mask in force.
    Also moves the Winograd threshold (WINO_MIN_BLOCKS) unless DT_CONV_WINO_MIN_BLOCKS pins it."""
    global WINO_MIN_BLOCKS, _PLAN_MASK
    got = _PLAN_MASK = int(_abi.lib().dt_conv_set_plan_objective(int(mask)))
    if "DT_CONV_WINO_MIN_BLOCKS" not in _os.environ:
        WINO_MIN_BLOCKS = _WINO_MIN_BLOCKS_BY_PLAN[bool(got & 1)]
    return got


def launch_config():
    """Hashable of everything process-wide that decides WHICH kernels a conv / volume call launches and with what grid: the
    module switches above and the library's settings token (plan objective, volume-kernel CU budget).  Replay mechanisms
    (utils/graphs.py, utils/program.py) bake those decisions in; the model keys its captured graphs / recorded programs on
    this, so a switch flipped afterwards leads to a new capture instead of a silent replay of the old choice."""
    return (int(_abi.lib().dt_settings_token()), WINO_MIN_BLOCKS, WINO4_MIN_BLOCKS, CONV_PRECISION, SPLIT_MIN_BLOCKS, TRANSPOSED_TILING, PAIR_LAUNCH,
            HEAD_MULTI_LAUNCH, HEADS_IN_CONV)


def _dev_param(conv, name, device):
    """fp32 device copy of a small parameter (bias / head weight), cached per version."""
    p = getattr(conv, name)
    if p is None:
        return None
    if p.device == device and p.dtype == torch.float32 and p.is_contiguous():
        return p  # (only its address is used)
    key = (p.data_ptr(), p._version, device)
    cache = conv.__dict__.setdefault("_dt_small", {})
    hit = cache.get(name)
    if hit is not None and hit[0] == key:
        _abi.wait_ready(hit[2], device)
        return hit[1]
    val = p.detach().to(device=device, dtype=torch.float32).contiguous()  # (a device-side cast/copy when p is on a GPU)
    cache[name] = (key, val, _abi.record_ready(device))
    return val


def _conv_plan(srcs, conv, act, impl, L):
    """Everything about a conv2d call that depends only on the conv module and the sources' shapes / strides / dtypes --
    validation, the launch descriptor, the kernel choice, the FLOP count -- computed once and cached on the module (the host
    enqueues ~45 conv launches per keyframe; with 4 keyframes in flight it needs 85 % of a step for that, so the per-call
    Python work is kept to key construction, one allocation and the ctypes call).  The key carries the strides and dtypes, so a
    hit implies the same NHWC fp32 layout that was validated on the miss."""
    key = (act, impl, WINO_MIN_BLOCKS, WINO4_MIN_BLOCKS, CONV_PRECISION, tuple((t.shape, t.stride(), t.dtype, up) for t, up in srcs))
    cache = conv.__dict__.get("_dt_plans")
    if cache is None:
        cache = conv.__dict__["_dt_plans"] = {}
    hit = cache.get(key)
    if hit is not None:
        return hit
    x0, up0 = srcs[0]
    n = x0.shape[0]
    h_in = x0.shape[2] * (2 if up0 else 1)
    w_in = x0.shape[3] * (2 if up0 else 1)
    k = conv.kernel_size[0]
    st = conv.stride[0]
    co = conv.out_channels
    d = _abi.ConvDesc()
    d.n, d.c_out, d.nsrc, d.ksize, d.stride, d.act = n, co, len(srcs), k, st, act
    d.h_in, d.w_in = h_in, w_in
    d.pad_mode = 1 if conv.padding_mode == "replicate" else 0
    pad = k // 2
    d.h_out = (h_in + 2 * pad - k) // st + 1
    d.w_out = (w_in + 2 * pad - k) // st + 1
    ctot = 0
    for i, (t, up) in enumerate(srcs):
        if not _is_nhwc(t):
            raise ValueError("conv2d sources must be NHWC fp32 (use as_nhwc)")
        hh, ww = t.shape[2] * (2 if up else 1), t.shape[3] * (2 if up else 1)
        if (hh, ww) != (h_in, w_in) or t.shape[0] != n:
            raise ValueError(f"source {i} extent {(hh, ww)} does not match {(h_in, w_in)}")
        d.c[i] = t.shape[1]
        d.up[i] = 1 if up else 0
        ctot += t.shape[1]
    if ctot != conv.in_channels:
        raise ValueError(f"conv expects {conv.in_channels} input channels, sources provide {ctot}")
    if impl == "mfma" and (co % 32 != 0 or any(d.c[i] % 8 != 0 for i in range(len(srcs)))):
        # channel counts the 32-channel x 8-channel-group MFMA tiling cannot express (the reference's own
        # configurations never produce them): the general-shape kernel, same fused epilogue
        impl = "simple"
    wino_blocks = n * ((d.h_out + 7) // 8) * ((d.w_out + 15) // 16) * (co // 32) if (k == 3 and st == 1 and co % 32 == 0) else 0
    if impl == "mfma" and wino_blocks > 0 and WINO_MIN_BLOCKS > 0 and wino_blocks >= WINO_MIN_BLOCKS:
        impl = "wino"
    if impl == "wino" and CONV_PRECISION == "fp32" and WINO4_MIN_BLOCKS > 0 and conv.padding_mode in ("zeros", "replicate") \
            and int(L.dt_conv2d_wino4_blocks(C.byref(d))) >= WINO4_MIN_BLOCKS:
        impl = "wino4"
    if impl == "wino" and CONV_PRECISION == "split16" and wino_blocks >= SPLIT_MIN_BLOCKS \
            and conv.padding_mode in ("zeros", "replicate") and L.dt_conv2d_wino_split_supported(C.byref(d)):
        impl = "wino_split"
    if impl == "mfma":
        d.transposed = 1 if _want_transposed(L, d) else 0
    # (the descriptor itself, not a byref object: ctypes structures survive copy.deepcopy / pickling of the module, references do not)
    plan = (d, impl, (n, co, d.h_out, d.w_out), 2.0 * n * d.h_out * d.w_out * co * ctot * k * k)
    if len(cache) >= 64:  # (varying input shapes: bounded; a fixed-shape loop uses one or two entries)
        cache.clear()
    cache[key] = plan
    return plan


def conv2d(srcs, conv: nn.Conv2d, act=ACT_NONE, residual=None, impl="mfma"):
    """Fused conv on NHWC tensors.

    srcs: list of (tensor [n,c,h,w] channels_last, upsample_nearest_x2: bool), concatenated
    along channels in order.  Returns a channels_last tensor [n, c_out, h_out, w_out].
    """
    L = _abi.lib()
    d, impl, oshape, flops = _conv_plan(srcs, conv, act, impl, L)
    dref = C.byref(d)
    dev = srcs[0][0].device
    nsrc = len(srcs)
    p0 = srcs[0][0].data_ptr()
    p1 = srcs[1][0].data_ptr() if nsrc > 1 else None
    p2 = srcs[2][0].data_ptr() if nsrc > 2 else None
    out = torch.empty(oshape, device=dev, dtype=torch.float32, memory_format=torch.channels_last)
    bias = _dev_param(conv, "bias", dev)
    pres = None
    if residual is not None:
        if not _is_nhwc(residual) or residual.shape != out.shape:
            raise ValueError("residual must be NHWC with the output's shape")
        pres = residual.data_ptr()
    pbias = None if bias is None else bias.data_ptr()
    stream = _abi.current_stream(dev)
    if ACCOUNT is not None:
        _account(flops)
    if impl == "wino":
        wp = packed_weight_wino(conv, dev)
        rc = L.dt_conv2d_wino_f32(dref, p0, p1, p2, wp.data_ptr(), pbias, pres, out.data_ptr(), stream)
        if rc:
            _abi.check(rc, "dt_conv2d_wino_f32")
    elif impl == "wino4":
        wp = packed_weight_wino4(conv, dev)
        rc = L.dt_conv2d_wino4_f32(dref, p0, p1, p2, wp.data_ptr(), pbias, pres, out.data_ptr(), stream)
        if rc:
            _abi.check(rc, "dt_conv2d_wino4_f32")
    elif impl == "mfma":
        wp = packed_weight(conv, dev, transposed=bool(d.transposed))
        rc = L.dt_conv2d_f32(dref, p0, p1, p2, wp.data_ptr(), pbias, pres, out.data_ptr(), stream)
        if rc:
            _abi.check(rc, "dt_conv2d_f32")
    elif impl == "wino_split":
        wp = packed_weight_wino_split(conv, dev)
        _abi.check(L.dt_conv2d_wino_split_f32(dref, p0, p1, p2, wp.data_ptr(), pbias, pres, out.data_ptr(), stream),
                   "dt_conv2d_wino_split_f32")
    else:
        wd = _dev_param(conv, "weight", dev)
        _abi.check(L.dt_conv2d_simple_f32(dref, p0, p1, p2, wd.data_ptr(), pbias, pres, out.data_ptr(), stream),
                   "dt_conv2d_simple_f32")
    return out


def _fill_desc(srcs, conv, act):
    x0, up0 = srcs[0]
    n = x0.shape[0]
    h_in = x0.shape[2] * (2 if up0 else 1)
    w_in = x0.shape[3] * (2 if up0 else 1)
    k, st = conv.kernel_size[0], conv.stride[0]
    d = _abi.ConvDesc()
    d.n, d.c_out, d.nsrc, d.ksize, d.stride, d.act = n, conv.out_channels, len(srcs), k, st, act
    d.h_in, d.w_in = h_in, w_in
    d.pad_mode = 1 if conv.padding_mode == "replicate" else 0
    pad = k // 2
    d.h_out = (h_in + 2 * pad - k) // st + 1
    d.w_out = (w_in + 2 * pad - k) // st + 1
    for i, (t, up) in enumerate(srcs):
        d.c[i] = t.shape[1]
        d.up[i] = 1 if up else 0
    return d


#: launch conv1 and the shortcut conv of a BasicBlock as one kernel (DT_CONV_PAIR=0 restores two launches; A/B switch)
PAIR_LAUNCH = _os.environ.get("DT_CONV_PAIR", "1") != "0"


def conv2d_pair(srcs, conv_a: nn.Conv2d, act_a, conv_b: nn.Conv2d, act_b):
    """Two convolutions of the same (virtually concatenated) sources in one launch (dt_conv2d_pair_f32): the 3x3 ``conv_a``
    and the shortcut ``conv_b`` (1x1 stride 1, or 3x3 stride 2) of a BasicBlock.  Returns (out_a, out_b): the same products
    as two conv2d calls, possibly in a different fp32 summation order (the pair plans its K splits jointly and may pick
    another split width than a lone launch would: differences of a few 1e-6 on O(1) maps, see
    tests/test_networks_gpu.py::test_paired_conv_launch_equals_two_launches).  Shapes the paired kernel cannot take go
    through two conv2d calls."""
    co_a, co_b = conv_a.out_channels, conv_b.out_channels
    ka, sa, kb, sb = conv_a.kernel_size[0], conv_a.stride[0], conv_b.kernel_size[0], conv_b.stride[0]
    ok = (PAIR_LAUNCH and ka == 3 and sa == sb and co_a % 32 == 0 and co_b % 32 == 0 and all(t.shape[1] % 8 == 0 for t, _ in srcs)
          and ((kb == 1 and sb == 1) or (kb == 3 and sb == 2)) and conv_a.padding_mode == "zeros" and conv_b.padding_mode == "zeros"
          and all(_is_nhwc(t) for t, _ in srcs))
    if not ok:
        return conv2d(srcs, conv_a, act=act_a), conv2d(srcs, conv_b, act=act_b)
    L = _abi.lib()
    dev = srcs[0][0].device
    da, db = _fill_desc(srcs, conv_a, act_a), _fill_desc(srcs, conv_b, act_b)
    ptrs = [_abi.ptr(t) for t, _ in srcs] + [None] * (3 - len(srcs))
    # same Winograd-vs-direct choice as conv2d makes for conv_a
    a_wino = False
    if sa == 1 and WINO_MIN_BLOCKS > 0:
        a_wino = da.n * ((da.h_out + 7) // 8) * ((da.w_out + 15) // 16) * (co_a // 32) >= WINO_MIN_BLOCKS
    if a_wino and CONV_PRECISION == "split16":
        # the split-precision Winograd kernel has no paired form: two launches
        return conv2d(srcs, conv_a, act=act_a), conv2d(srcs, conv_b, act=act_b)
    if not a_wino:
        da.transposed = 1 if _want_transposed(L, da) else 0
    db.transposed = 1 if _want_transposed(L, db) else 0
    wa = packed_weight_wino(conv_a, dev) if a_wino else packed_weight(conv_a, dev, transposed=bool(da.transposed))
    wb = packed_weight(conv_b, dev, transposed=bool(db.transposed))
    out_a = empty_nhwc(da.n, co_a, da.h_out, da.w_out, dev)
    out_b = empty_nhwc(db.n, co_b, db.h_out, db.w_out, dev)
    cin = sum(t.shape[1] for t, _ in srcs)
    _account(2.0 * da.n * da.h_out * da.w_out * cin * (co_a * ka * ka + co_b * kb * kb))
    _abi.check(L.dt_conv2d_pair_f32(C.byref(da), C.byref(db), ptrs[0], ptrs[1], ptrs[2], _abi.ptr(wa), int(a_wino),
                                    _abi.ptr(_dev_param(conv_a, "bias", dev)), _abi.ptr(out_a), _abi.ptr(wb),
                                    _abi.ptr(_dev_param(conv_b, "bias", dev)), _abi.ptr(out_b), _abi.current_stream(dev)),
               "dt_conv2d_pair_f32")
    return out_a, out_b


def conv1x1_head(x, conv: nn.Conv2d, with_exp=False):
    """1x1 conv to a single channel (regression head): NHWC [n,c,h,w] -> [n,1,h,w].
    with_exp: return (out, exp(out)) from the same launch."""
    if conv.kernel_size != (1, 1) or conv.out_channels != 1:
        raise NotImplementedError("head must be a 1x1 conv to one channel")
    L = _abi.lib()
    n, c, h, w = x.shape
    dev = x.device
    out = torch.empty((n, 1, h, w), device=dev, dtype=torch.float32)
    wv = _dev_param(conv, "weight", dev)
    b = _dev_param(conv, "bias", dev)
    out_e = torch.empty_like(out) if with_exp else None
    _account(2.0 * n * h * w * c)
    _abi.check(L.dt_conv1x1_head_f32(_abi.ptr(x), _abi.ptr(wv), _abi.ptr(b), _abi.ptr(out), _abi.ptr(out_e), n * h * w, c,
                                     _abi.current_stream(dev)), "dt_conv1x1_head_f32")
    return (out, out_e) if with_exp else out


def _head_pack(head: nn.Sequential, dev):
    from . import mlp_pack

    ca, cb_, cc = head[0], head[2], head[4]
    params = [ca.weight, ca.bias, cb_.weight, cb_.bias, cc.weight, cc.bias]
    key = (dev,) + tuple((p.data_ptr(), p._version) for p in params)
    hit = head.__dict__.get("_dt_head_pack")
    if hit is None or hit[0] != key:
        arrs = [p.detach().float().cpu().numpy() for p in params]
        pk = mlp_pack.pack_head_mlp(*arrs)
        # (blocking host-to-device copies: complete on return, but the token keeps every cache uniform)
        hit = (key, {k: torch.from_numpy(v).to(dev) for k, v in pk.items()}, _abi.record_ready(dev))
        head.__dict__["_dt_head_pack"] = hit
    else:
        _abi.wait_ready(hit[2], dev)
    return hit[1]


HEAD_MULTI_MAX_PIXELS = 32 * 1024  # per map; larger maps take the persistent kernel of dt_head_mlp_f32
HEAD_MULTI_LAUNCH = _os.environ.get("DT_HEAD_MULTI", "1") != "0"


def head_mlp_multi(xs, heads, with_exp=False):
    """head_mlp for 2-4 independent small feature maps (the coarse decoder scales) in ONE launch; returns a list
    like [head_mlp(x, h, with_exp) for x, h in zip(xs, heads)], bit-identical to it."""
    L = _abi.lib()
    n_h = len(xs)
    dev = xs[0].device
    pks = [_head_pack(h, dev) for h in heads]
    outs = [torch.empty((x.shape[0], 1, x.shape[2], x.shape[3]), device=dev, dtype=torch.float32) for x in xs]
    outs_e = [torch.empty_like(o) for o in outs] if with_exp else [None] * n_h
    tab = lambda vals: (C.c_void_p * n_h)(*[_abi.ptr(v) for v in vals])
    pixels = (C.c_int64 * n_h)(*[x.shape[0] * x.shape[2] * x.shape[3] for x in xs])
    cin = (C.c_int * n_h)(*[x.shape[1] for x in xs])
    _account(sum(2.0 * x.shape[0] * x.shape[2] * x.shape[3] * (x.shape[1] * 128 + 128 * 128 + 128) for x in xs))
    _abi.check(L.dt_head_mlp_multi_f32(n_h, tab(xs), tab([p["wa"] for p in pks]), tab([p["wb"] for p in pks]),
                                       tab([p["tail"] for p in pks]), tab(outs), tab(outs_e), pixels, cin,
                                       _abi.current_stream(dev)), "dt_head_mlp_multi_f32")
    return [(o, e) for o, e in zip(outs, outs_e)] if with_exp else outs


#: run the coarse regression heads as workgroups of the last decoder block's first convolution (dt_conv2d_wino_heads_f32;
#: DT_HEADS_IN_CONV=0 restores the two launches: A/B switch, results are bit-identical either way)
HEADS_IN_CONV = _os.environ.get("DT_HEADS_IN_CONV", "1") != "0"


def conv2d_with_heads(srcs, conv: nn.Conv2d, act, xs, heads, with_exp=False):
    """``conv2d(srcs, conv, act)`` and ``head_mlp_multi(xs, heads, with_exp)`` in ONE launch when the convolution is a
    chip-filling Winograd layer (otherwise, and for shapes the fused entry does not take, the two calls).  Returns
    (conv output, list of head results), bit-identical to the two calls."""
    L = _abi.lib()
    d, impl, oshape, flops = _conv_plan(srcs, conv, act, "mfma", L)
    dref = C.byref(d)
    if not (HEADS_IN_CONV and HEAD_MULTI_LAUNCH and impl == "wino" and 1 <= len(xs) <= 4):
        return conv2d(srcs, conv, act=act), (head_mlp_multi(xs, heads, with_exp=with_exp) if (len(xs) >= 2 and HEAD_MULTI_LAUNCH) else
                                              [head_mlp(x, h, with_exp=with_exp) for x, h in zip(xs, heads)])
    dev = srcs[0][0].device
    nsrc = len(srcs)
    p0 = srcs[0][0].data_ptr()
    p1 = srcs[1][0].data_ptr() if nsrc > 1 else None
    p2 = srcs[2][0].data_ptr() if nsrc > 2 else None
    out = torch.empty(oshape, device=dev, dtype=torch.float32, memory_format=torch.channels_last)
    bias = _dev_param(conv, "bias", dev)
    n_h = len(xs)
    pks = [_head_pack(h, dev) for h in heads]
    outs = [torch.empty((x.shape[0], 1, x.shape[2], x.shape[3]), device=dev, dtype=torch.float32) for x in xs]
    outs_e = [torch.empty_like(o) for o in outs] if with_exp else [None] * n_h
    tab = lambda vals: (C.c_void_p * n_h)(*[_abi.ptr(v) for v in vals])
    pixels = (C.c_int64 * n_h)(*[x.shape[0] * x.shape[2] * x.shape[3] for x in xs])
    cin = (C.c_int * n_h)(*[x.shape[1] for x in xs])
    if ACCOUNT is not None:
        _account(flops)
        _account(sum(2.0 * x.shape[0] * x.shape[2] * x.shape[3] * (x.shape[1] * 128 + 128 * 128 + 128) for x in xs))
    wp = packed_weight_wino(conv, dev)
    _abi.check(L.dt_conv2d_wino_heads_f32(dref, p0, p1, p2, wp.data_ptr(), None if bias is None else bias.data_ptr(), None,
                                          out.data_ptr(), n_h, tab(xs), tab([p["wa"] for p in pks]), tab([p["wb"] for p in pks]),
                                          tab([p["tail"] for p in pks]), tab(outs), tab(outs_e), pixels, cin,
                                          _abi.current_stream(dev)), "dt_conv2d_wino_heads_f32")
    return out, ([(o, e) for o, e in zip(outs, outs_e)] if with_exp else outs)


def head_mlp(x, head: nn.Sequential, with_exp=False):
    """Fused 1x1 -> ELU -> 1x1 -> ELU -> 1x1 regression head (modules/networks_fast.py:102-132).
    x NHWC [n,c,h,w] (c = 64, 128, or 256 for small maps) -> [n,1,h,w]; with_exp: (out, exp(out)) from the same launch."""
    L = _abi.lib()
    n, c, h, w = x.shape
    dev = x.device
    pk = _head_pack(head, dev)
    out = torch.empty((n, 1, h, w), device=dev, dtype=torch.float32)
    out_e = torch.empty_like(out) if with_exp else None
    _account(2.0 * n * h * w * (c * 128 + 128 * 128 + 128))
    _abi.check(L.dt_head_mlp_f32(_abi.ptr(x), _abi.ptr(pk["wa"]), _abi.ptr(pk["wb"]), _abi.ptr(pk["tail"]), _abi.ptr(out),
                                 _abi.ptr(out_e), n * h * w, c, _abi.current_stream(dev)), "dt_head_mlp_f32")
    return (out, out_e) if with_exp else out


def head_mlp_supported(x, head) -> bool:
    try:
        cin_ok = x.shape[1] in (64, 128) or (x.shape[1] == 256 and x.shape[0] * x.shape[2] * x.shape[3] <= 32 * 1024)
        return (cin_ok and len(head) == 5 and head[0].out_channels == 128 and head[2].out_channels == 128
                and head[4].out_channels == 1 and all(head[i].kernel_size == (1, 1) for i in (0, 2, 4)))
    except Exception:
        return False


def upsample2x_bilinear(x):
    """utils/generic_utils.py:95-104 on an NHWC tensor."""
    L = _abi.lib()
    n, c, h, w = x.shape
    out = empty_nhwc(n, c, 2 * h, 2 * w, x.device)
    _abi.check(L.dt_upsample2x_bilinear_f32(_abi.ptr(x), _abi.ptr(out), n, h, w, c, _abi.current_stream(x.device)),
               "dt_upsample2x_bilinear_f32")
    return out


def exp(x):
    L = _abi.lib()
    src = x if x.is_contiguous() else x.contiguous()
    out = torch.empty_like(src)
    _abi.check(L.dt_exp_f32(_abi.ptr(src), _abi.ptr(out), src.numel(), _abi.current_stream(x.device)), "dt_exp_f32")
    return out
