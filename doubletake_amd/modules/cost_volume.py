"""Drop-in plane-sweep cost-volume managers backed by the gfx950 HIP kernels.

Same class names, constructor and forward signatures, return tuples and state-dict keys as the
reference (paths relative to /root/reference/src/doubletake/):

    CostVolumeManager                 modules/cost_volume.py:9-363
    FeatureVolumeManager              modules/feature_volume.py:12-365  (+ Fast :368-796)
    FeatureMeshHintVolumeManager      modules/mesh_hint_volume.py:12-449 (+ Fast :452-928)

Plug point: ``model.cost_volume = <one of these>`` exactly as the reference swaps in its own
fast path (utils/model_utils.py:30-35).  ``load_state_dict`` of a reference checkpoint's
``cost_volume.*`` keys works unchanged.

Differences kept on purpose (SURVEY.md "facts" 6): geometry is derived from the input shape on
every call, so any (h, w) works, landscape or portrait.  Shapes outside the tuned kernels --
``depth_planes_bdhw`` that vary over the image, ``matching_dim_size`` != 16 (<= 32), more than 15
source views (<= 16) -- run on the general one-thread-per-sample HIP kernels (same results, not the
tuned path; no released model uses them).

There is no torch fallback: forward() raises if the tensors are not on a GPU or the HIP
library cannot be loaded.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import os

import torch
import torch.nn as nn

from .. import _abi
from ..utils import graphs as _graphs
from . import mlp_pack
from .networks import MLP


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _abi.DoubletakeHipError(
            f"{name} is on {t.device}; the doubletake_amd cost volume only runs on a ROCm GPU (no CPU fallback)"
        )


class _Buffers(nn.Module):
    """Holds the reference's geometry buffers so checkpoints load with strict=True."""

    def __init__(self, **bufs):
        super().__init__()
        for k, v in bufs.items():
            self.register_buffer(k, v)


class CostVolumeManager(nn.Module):
    """Dot-product plane-sweep volume (reference modules/cost_volume.py:9)."""

    #: write the volume as torch.channels_last ([b,D,h,w] logical, NHWC physical); the conv
    #: stack that consumes it (CVEncoder) is NHWC.  Only honoured by the MLP volumes.
    channels_last_output = True
    #: optional callable(tag) invoked right before / after dt_cv_dot_f32 (bench.py: HIP events)
    _dot_event_hook = None
    #: "lds" (product path) | "direct" | "stats" -- see forward()
    _dot_impl = "lds"

    def __init__(self, matching_height, matching_width, num_depth_bins=64, matching_dim_size=None,
                 num_source_views=None):
        super().__init__()
        self.num_depth_bins = num_depth_bins
        self.matching_height = matching_height
        self.matching_width = matching_width
        self._planes_px = None  # caller-supplied per-pixel planes of the call in flight (set by _setup)
        self.initialise_for_projection()

    # -- reference API ---------------------------------------------------------------------
    def initialise_for_projection(self, device=None):
        """Buffers with the reference's names/shapes (cost_volume.py:51-71,
        geometry_utils.py:28-52,71-75).  The kernels do not read them."""
        ramp = torch.linspace(0, 1, self.num_depth_bins).view(1, self.num_depth_bins, 1, 1)
        self.register_buffer("linear_ramp_1d11", ramp)
        h, w = self.matching_height, self.matching_width
        xx, yy = torch.meshgrid(torch.arange(w), torch.arange(h), indexing="xy")
        pix = torch.stack((xx, yy), 0) + 0.5
        pix = torch.cat([pix, torch.ones_like(pix[:1])], 0).flatten(1).unsqueeze(0)
        self.backprojector = _Buffers(pix_coords_13N=pix)
        self.projector = _Buffers(eps=torch.tensor(1e-8).view(1, 1, 1))
        if device is not None:
            self.to(device)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # the pixel grid is shape dependent and unused here: never fail on a size mismatch
        key = prefix + "backprojector.pix_coords_13N"
        if key in state_dict and state_dict[key].shape != self.backprojector.pix_coords_13N.shape:
            state_dict = dict(state_dict)
            state_dict[key] = self.backprojector.pix_coords_13N
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def get_mask(self, pix_coords_bk2hw):
        """cost_volume.py:73-94 (torch, for API parity; the kernels have their own)."""
        u, v = pix_coords_bk2hw[:, :, 0], pix_coords_bk2hw[:, :, 1]
        return (u > 2) & (u < self.matching_width - 2) & (v > 2) & (v < self.matching_height - 2)

    @torch.no_grad()
    def generate_depth_planes(self, batch_size, min_depth, max_depth):
        """cost_volume.py:96-130: log-spaced planes [batch, D, h, w] (an expanded view).  Computed by the same
        dt_cv_setup_f32 launch forward() uses (with identity cameras), so the values are bit-identical to the
        planes the volume kernels see."""
        _require_gpu(min_depth, "min_depth")
        L = _abi.lib()
        dev = min_depth.device
        b, D = int(batch_size), self.num_depth_bins
        eye = torch.eye(4, device=dev, dtype=torch.float32)
        m1 = eye.repeat(b, 1, 1, 1).contiguous()
        invK = eye.repeat(b, 1, 1).contiguous()
        mn = _f32c(min_depth).reshape(-1)
        mx = _f32c(max_depth.to(dev)).reshape(-1)
        mn = mn if mn.numel() == b else mn[:1].expand(b).contiguous()
        mx = mx if mx.numel() == b else mx[:1].expand(b).contiguous()
        params = torch.empty(b, int(L.dt_cv_params_floats(D, 1)), device=dev, dtype=torch.float32)
        _abi.check(L.dt_cv_setup_f32(_abi.ptr(m1), _abi.ptr(m1), _abi.ptr(m1), _abi.ptr(invK), _abi.ptr(mn), _abi.ptr(mx),
                                     b, 1, D, _abi.ptr(params), _abi.current_stream(dev)), "dt_cv_setup_f32")
        return params[:, 12:12 + D].reshape(b, D, 1, 1).expand(b, D, self.matching_height, self.matching_width)

    @torch.no_grad()
    def warp_features(self, src_feats, src_extrinsics, src_Ks, cur_invK, depth_plane_b1hw, batch_size, num_src_frames,
                      num_feat_channels, uv_scale=None):
        """cost_volume.py:132-217 -> (world_points_B4N, depths [b,k,h,w], src_feat_warped [b,k,c,h,w], mask).
        Stand-alone form of the warp the fused volume kernels do internally (``uv_scale`` is implied by the
        matching resolution)."""
        _require_gpu(src_feats, "src_feats")
        L = _abi.lib()
        dev = src_feats.device
        b, k, c = int(batch_size), int(num_src_frames), int(num_feat_channels)
        h, w = self.matching_height, self.matching_width
        src = _f32c(src_feats).view(b, k, c, h, w)
        Ks = _f32c(src_Ks).view(b, k, 4, 4)
        ext = _f32c(src_extrinsics).view(b, k, 4, 4)
        invK = _f32c(cur_invK).view(b, 4, 4)
        one = torch.ones(b, device=dev, dtype=torch.float32)
        params = torch.empty(b, int(L.dt_cv_params_floats(1, k)), device=dev, dtype=torch.float32)
        stream = _abi.current_stream(dev)
        _abi.check(L.dt_cv_setup_f32(_abi.ptr(Ks), _abi.ptr(ext), _abi.ptr(ext), _abi.ptr(invK), _abi.ptr(one), _abi.ptr(one),
                                     b, k, 1, _abi.ptr(params), stream), "dt_cv_setup_f32")
        depth = _f32c(depth_plane_b1hw).view(b, h, w)
        world = torch.empty(b * k, 4, h * w, device=dev, dtype=torch.float32)
        depths = torch.empty(b, k, h, w, device=dev, dtype=torch.float32)
        warped = torch.empty(b, k, c, h, w, device=dev, dtype=torch.float32)
        mask = torch.empty(b, k, h, w, device=dev, dtype=torch.float32)
        _abi.check(L.dt_cv_warp_f32(_abi.ptr(src), _abi.ptr(params), _abi.ptr(depth), b, k, c, h, w, 1, _abi.ptr(world),
                                    _abi.ptr(depths), _abi.ptr(warped), _abi.ptr(mask), stream), "dt_cv_warp_f32")
        return world, depths, warped, mask

    def indices_to_disparity(self, indices, depth_planes_bdhw):
        return torch.gather(depth_planes_bdhw, 1, indices.unsqueeze(1)).squeeze(1)

    # -- shared plumbing ---------------------------------------------------------------------
    def _per_batch(self, t, b, dev, slot):
        """min / max depth as a dense fp32 [b] device tensor.  The reference passes [1,1,1,1] tensors for any batch size: the
        expanded copy is kept per (tensor version, b), so a fixed-shape loop launches no torch kernel here (a launch program,
        utils/program.py, could not replay one)."""
        flat = t.reshape(-1)
        if flat.numel() == b and flat.device == dev and flat.dtype == torch.float32 and flat.is_contiguous():
            return flat
        key = (t.data_ptr(), t._version, b, str(dev))
        hit = self.__dict__.get(slot)
        if hit is not None and hit[0] == key:
            _abi.wait_ready(hit[2], dev)
            return hit[1]
        v = _f32c(t.to(dev)).reshape(-1)
        if v.numel() != b:
            v = v.expand(b).contiguous()
        self.__dict__[slot] = (key, v, _abi.record_ready(dev))
        return v

    def _setup(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
               depth_planes_bdhw):
        _require_gpu(cur_feats, "cur_feats")
        L = _abi.lib()
        b, k, c, h, w = src_feats.shape
        if cur_feats.shape != (b, c, h, w):
            raise ValueError(f"cur_feats {tuple(cur_feats.shape)} does not match src_feats {tuple(src_feats.shape)}")
        self.matching_height, self.matching_width = h, w
        D = self.num_depth_bins
        dev = cur_feats.device
        stream = _abi.current_stream(dev)
        cur = _f32c(cur_feats)
        src = _f32c(src_feats)
        Ks = _f32c(src_Ks).view(b, k, 4, 4)
        ext = _f32c(src_extrinsics).view(b, k, 4, 4)
        poses = _f32c(src_poses).view(b, k, 4, 4)
        invK = _f32c(cur_invK).view(b, 4, 4)
        mn = self._per_batch(min_depth, b, dev, "_dt_min_b")
        mx = self._per_batch(max_depth, b, dev, "_dt_max_b")
        pf = L.dt_cv_params_floats(D, k)
        params = torch.empty(b, pf, device=dev, dtype=torch.float32)
        _abi.check(L.dt_cv_setup_f32(_abi.ptr(Ks), _abi.ptr(ext), _abi.ptr(poses), _abi.ptr(invK), _abi.ptr(mn),
                                     _abi.ptr(mx), b, k, D, _abi.ptr(params), stream), "dt_cv_setup_f32")
        planes_px = None
        if depth_planes_bdhw is not None:
            dp = depth_planes_bdhw
            if dp.shape[0] != b or dp.shape[1] != D:
                raise ValueError("depth_planes_bdhw must be [b, num_depth_bins, h, w]")
            flat = dp.reshape(b, D, -1)
            if flat.shape[-1] == 1 or bool((flat == flat[:, :, :1]).all()):
                params[:, 12:12 + D] = flat[:, :, 0].float()   # one depth per plane: the tuned kernels take it as is
            elif flat.shape[-1] != h * w:
                raise ValueError("depth_planes_bdhw must be [b, num_depth_bins, h, w]")
            else:
                # planes that vary over the image (cost_volume.py:249-250 takes any [b,D,h,w]): the general
                # one-thread-per-sample kernels read the depth of every (plane, pixel) from this tensor
                planes_px = _f32c(dp.to(dev)).view(b, D, h, w)
        src_nhwc = torch.empty(b, k, h, w, c, device=dev, dtype=torch.float32)
        _abi.check(L.dt_nchw_to_nhwc_f32(_abi.ptr(src), _abi.ptr(src_nhwc), b * k, c, h, w, stream),
                   "dt_nchw_to_nhwc_f32")
        self._planes_px = planes_px
        planes_bdhw = planes_px if planes_px is not None else params[:, 12:12 + D].reshape(b, D, 1, 1).expand(b, D, h, w)
        return L, stream, cur, src_nhwc, params, planes_bdhw, (b, k, c, h, w, D)

    def _lowest(self, L, stream, vol, params, nhwc, dims):
        b, k, c, h, w, D = dims
        low = torch.empty(b, h, w, device=vol.device, dtype=torch.float32)
        _abi.check(L.dt_cv_lowest_cost_f32(_abi.ptr(vol), _abi.ptr(params), _abi.ptr(self._planes_px), _abi.ptr(low),
                                           int(nhwc), b, k, h, w, D, stream), "dt_cv_lowest_cost_f32")
        return low

    def _mask(self, L, stream, params, per_view, dims):
        b, k, c, h, w, D = dims
        shape = (b, k, h, w) if per_view else (b, h, w)
        m = torch.empty(shape, device=params.device, dtype=torch.uint8)
        _abi.check(L.dt_cv_overall_mask_u8(_abi.ptr(params), _abi.ptr(self._planes_px), _abi.ptr(m), int(per_view), b, k,
                                           h, w, D, stream), "dt_cv_overall_mask_u8")
        return m.view(torch.bool)  # bytes are 0/1: reinterpret, no copy kernel

    # -- forward -------------------------------------------------------------------------------
    def build_cost_volume(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth,
                          max_depth, depth_planes_bdhw=None, return_mask=False):
        """cost_volume.py:219-315 -> (cost_volume_bdhw, depth_planes_bdhw, None)."""
        vol, _, planes, _ = self.forward(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth,
                                         max_depth, depth_planes_bdhw, return_mask)
        return vol, planes, None

    @torch.no_grad()
    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        """cost_volume.py:322-363 -> (cost_volume, lowest_cost, depth_planes_bdhw, None)."""
        L, stream, cur, src_nhwc, params, planes, dims = self._setup(
            cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth, depth_planes_bdhw)
        b, k, c, h, w, D = dims
        vol = torch.empty(b, D, h, w, device=cur.device, dtype=torch.float32)
        hook = CostVolumeManager._dot_event_hook
        if hook is not None:
            hook("dot_begin")
        impl = CostVolumeManager._dot_impl
        if c != 16 or self._planes_px is not None:
            # shapes outside the tuned kernel (matching_feature_dims != 16, per-pixel planes): general kernel
            _abi.check(L.dt_cv_dot_simple_f32(_abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(self._planes_px),
                                              _abi.ptr(vol), b, k, c, h, w, D, stream), "dt_cv_dot_simple_f32")
        elif impl == "lds":      # source footprint of each pixel tile staged in LDS (csrc/cv_dot_lds.hip)
            _abi.check(L.dt_cv_dot_f32(_abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(vol), b, k, c, h, w, D,
                                       stream), "dt_cv_dot_f32")
        elif impl == "direct":  # every tap from global memory: same expressions, bit-identical (tests, ablation)
            _abi.check(L.dt_cv_dot_direct_f32(_abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(vol), b, k, c, h,
                                              w, D, stream), "dt_cv_dot_direct_f32")
        elif impl == "stats":
            self.last_dot_stats = torch.zeros(4, dtype=torch.int32, device=cur.device)
            _abi.check(L.dt_cv_dot_stats_f32(_abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(vol), b, k, c, h,
                                             w, D, _abi.ptr(self.last_dot_stats), stream), "dt_cv_dot_stats_f32")
        else:
            raise ValueError(impl)
        if hook is not None:
            hook("dot_end")
        low = self._lowest(L, stream, vol, params, False, dims)
        return vol, low, planes, None


class FeatureVolumeManager(CostVolumeManager):
    """Metadata-MLP volume (SimpleRecon; reference modules/feature_volume.py:12)."""

    _has_hint = False
    _fast_mask = True  # feature_volume.py:250-259: even the loop version returns any_k masks
    #: optional callable(tag) invoked right before / after the fused kernel launch; bench.py uses
    #: it to record HIP events on the launch stream (roofline timing)
    _event_hook = None
    #: source views the fused MFMA kernel handles (84 KB of layer-1 weights per 7 views stay in LDS)
    #: feature channels the general kernel takes (the fused kernel is built for the reference default, 16: options.py:127-136)
    MAX_GENERAL_CHANNELS = 32
    MAX_FUSED_VIEWS = 15       # fused MFMA kernel: <= 7 views fully LDS-resident, 8..15 with the further views streamed from L2
    MAX_SPLIT16_VIEWS = 7      # the opt-in split-precision kernel keeps every view resident
    _warned_views = False
    #: arithmetic of the fused MLP volume: "fp32" (default, exact fp32 MFMA) or "split16" (OPT-IN: fp16 hi/lo operands on
    #: the fp16 matrix pipe, fp32-class accuracy, ~4x less matrix time; csrc/cv_mlp_split.hip).  Set on an instance
    #: (`manager.precision = "split16"`) or for the process with DT_MLP_PRECISION=split16.
    precision = os.environ.get("DT_MLP_PRECISION", "fp32")
    #: cost-aware span plan in front of the fused hint kernel (dt_cv_mlp_plan_f32 + dt_cv_mlp_hint_planned_f32): two small
    #: launches that give every wave a span of equal estimated work; same volume (the plan only moves span boundaries).  DT_MLP_PLAN=0 switches it off in the library as well.
    use_span_plan = os.environ.get("DT_MLP_PLAN", "1") != "0"

    def __init__(self, matching_height, matching_width, num_depth_bins=64, mlp_channels=None, matching_dim_size=16,
                 num_source_views=7):
        super().__init__(matching_height, matching_width, num_depth_bins)
        mlp_channels = list(mlp_channels) if mlp_channels is not None else [202, 128, 128, 1]
        if not 1 <= int(matching_dim_size) <= self.MAX_GENERAL_CHANNELS:
            raise ValueError(f"matching_dim_size={matching_dim_size} not in 1..{self.MAX_GENERAL_CHANNELS}")
        self.matching_dim_size = int(matching_dim_size)
        self.num_source_views = num_source_views
        # feature_volume.py:49-67: visual (1+K)*C + depth (1+K) + rays 3(1+K) + angle K + mask K + dot K + pose 3K
        mlp_channels[0] = (self.matching_dim_size + 4) * (num_source_views + 1) + 6 * num_source_views
        assert self.matching_dim_size != 16 or mlp_channels[0] == mlp_pack.Columns(num_source_views).total
        if mlp_channels[1:] != [128, 128, 1]:
            raise NotImplementedError("the fused kernel implements the reference's [Cin,128,128,1] matching MLP")
        self.mlp = MLP(channel_list=mlp_channels, disable_final_activation=True)
        if self._has_hint:
            self.hint_mlp = MLP(channel_list=[3, 12, 12, 1], disable_final_activation=True)
        self._pack_cache = {}

    # -- weights -----------------------------------------------------------------------------
    def _mlp_arrays(self, mlp):
        lin = [m for m in mlp.net if isinstance(m, nn.Linear)]
        return [a for l in lin for a in (l.weight.detach(), l.bias.detach())]

    def _packed(self, device):
        params = list(self.mlp.parameters()) + (list(self.hint_mlp.parameters()) if self._has_hint else [])
        key = (str(device), self.precision) + tuple((p.data_ptr(), p._version) for p in params)
        hit = self._pack_cache.get("key")
        if hit == key:
            _abi.wait_ready(self._pack_cache["ready"], device)
            return self._pack_cache["val"]
        arrs = [a.float().cpu().numpy() for a in self._mlp_arrays(self.mlp)]
        val = {}
        if self.num_source_views <= self.MAX_FUSED_VIEWS and self.matching_dim_size == 16:
            # (the metadata-step layout is a build property of the library: ask it how many layer-1 floats it expects)
            n_dyn = C.c_int(0)
            _abi.check(_abi.lib().dt_cv_mlp_pack_floats(self.num_source_views, C.byref(n_dyn), None, None, None),
                       "dt_cv_mlp_pack_floats")
            paired = n_dyn.value == mlp_pack.dyn_steps_total(self.num_source_views, True) * 256
            if not paired and n_dyn.value != mlp_pack.dyn_steps_total(self.num_source_views, False) * 256:
                raise _abi.DoubletakeHipError(f"library expects {n_dyn.value} packed layer-1 floats: unknown step layout")
            packed = mlp_pack.pack_mlp(*arrs, self.num_source_views, paired=paired)
            val = {n: torch.from_numpy(v).to(device) for n, v in packed.items()}
        raw = self._mlp_arrays(self.mlp)
        val["raw"] = [_f32c(a.to(device)) for a in raw]
        if self.num_source_views <= self.MAX_SPLIT16_VIEWS and self.precision == "split16" and self.matching_dim_size == 16:
            sp = mlp_pack.pack_mlp_split(*arrs, self.num_source_views)
            for n in ("w1dyn", "w1pix", "w2"):
                val["sp_" + n] = torch.from_numpy(sp[n].view(np.int16).copy()).to(device)
        if self._has_hint:
            h = [a.float().cpu().numpy() for a in self._mlp_arrays(self.hint_mlp)]
            val["hint"] = torch.from_numpy(mlp_pack.pack_hint_mlp(*h)).to(device)
        self._pack_cache = {"key": key, "val": val, "ready": _abi.record_ready(device)}
        return val

    # -- forward -------------------------------------------------------------------------------
    def build_cost_volume(self, *args, **kwargs):
        vol, _, planes, mask = self.forward(*args, **kwargs)
        return vol, planes, mask

    @torch.no_grad()
    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        """feature_volume.py forward -> (cost_volume, lowest_cost, depth_planes_bdhw, overall_mask_bhw)."""
        return self._forward_impl(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth,
                                  max_depth, None, depth_planes_bdhw, return_mask)

    def _forward_impl(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                      cv_depth_hint_dict, depth_planes_bdhw, return_mask, _impl="mfma"):
        L, stream, cur, src_nhwc, params, planes, dims = self._setup(
            cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth, depth_planes_bdhw)
        b, k, c, h, w, D = dims
        if k != self.num_source_views:
            raise ValueError(f"built for {self.num_source_views} source views, got {k}")
        if c != self.matching_dim_size:
            raise ValueError(f"built for {self.matching_dim_size}-channel matching features, got {c}")
        dev = cur.device
        if (c != 16 or self._planes_px is not None) and _impl == "mfma":
            # matching_dim_size != 16 / planes that vary over the image: the reference accepts both
            # (mesh_hint_volume.py:32,95), no released model uses them -- general kernel, not the tuned path
            _impl = "simple"
        if k > self.MAX_FUSED_VIEWS and _impl == "mfma":
            # the fused MFMA kernel takes up to 15 source views (the reference's default is 7; the views beyond the seventh
            # stream their layer-1 weights from L2); more run on the general kernel -- correct, but not the tuned path
            if not FeatureVolumeManager._warned_views:
                import warnings

                warnings.warn(f"{k} source views: the fused MLP volume kernel covers up to {self.MAX_FUSED_VIEWS}; using the "
                              "general (one thread per pixel and plane) HIP kernel", stacklevel=2)
                FeatureVolumeManager._warned_views = True
            _impl = "simple"
        pk = self._packed(dev)
        hint_ptr = hd = hw_ = hm = None
        H2 = W2 = 0
        if self._has_hint:
            if cv_depth_hint_dict is None:
                raise ValueError("cv_depth_hint_dict is required (depth_hint_b1hw, sampled_weights_b1hw, depth_hint_mask_b1hw)")
            hd = _f32c(cv_depth_hint_dict["depth_hint_b1hw"].to(dev))
            hw_ = _f32c(cv_depth_hint_dict["sampled_weights_b1hw"].to(dev))
            hm = _f32c(cv_depth_hint_dict["depth_hint_mask_b1hw"].to(dev))
            if not (hd.shape == hw_.shape == hm.shape) or hd.shape[0] != b or hd.shape[1] != 1:
                raise ValueError("hint maps must all be [b,1,H,W]")
            H2, W2 = hd.shape[-2:]
            hint_ptr = pk["hint"]
        nhwc = bool(self.channels_last_output) and _impl == "mfma"
        if nhwc:
            vol = torch.empty(b, D, h, w, device=dev, dtype=torch.float32, memory_format=torch.channels_last)
        else:
            vol = torch.empty(b, D, h, w, device=dev, dtype=torch.float32)
        planned = (_impl == "mfma" and self._has_hint and self.use_span_plan
                   and not (self.precision == "split16" and k <= self.MAX_SPLIT16_VIEWS))
        plan = None
        if planned:
            # cost-aware span plan of the hint kernel (it skips views a tile cannot see: units differ in cost): two small launches
            # that read only the cameras and planes in `params`, in front of the event bracket of the volume kernel itself.  The
            # scratch is a fresh torch allocation on the current stream, like the volume
            plan = torch.empty(int(L.dt_cv_mlp_plan_bytes(b, h, w, D)), dtype=torch.uint8, device=dev)
            self._last_plan = plan  # (diagnostics / tests: [span bounds int32 | group totals uint32 | in-group price prefixes uint32])
            _abi.check(L.dt_cv_mlp_plan_f32(_abi.ptr(params), b, k, h, w, D, _abi.ptr(plan), int(plan.numel()), stream),
                       "dt_cv_mlp_plan_f32")
        hook = FeatureVolumeManager._event_hook
        if hook is not None or _graphs.recording() or "_stage_hook" in self.__dict__:
            # replay mechanisms: segment boundary so that the hook's events bracket the kernel on replay too.  hipGraph capture:
            # only with a hook installed (every cut is one more hipGraphLaunch per replay); launch programs: always (a segment
            # costs one more C call, and a hook installed later then needs no new recording)
            _graphs.cut("mlp_begin")
        gate = self.__dict__.get("_stage_hook")  # per-instance: parallel.KeyframePipeline orders the volume kernels of its lanes
        if not _graphs.building():
            if gate is not None:
                gate("mlp_begin")
            if hook is not None:
                hook("mlp_begin")
        if _impl == "mfma" and self.precision == "split16" and k <= self.MAX_SPLIT16_VIEWS:
            _abi.check(L.dt_cv_mlp_hint_split_f32(
                _abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(pk["sp_w1dyn"]), _abi.ptr(pk["sp_w1pix"]),
                _abi.ptr(pk["sp_w2"]), _abi.ptr(pk["tail"]), _abi.ptr(hint_ptr), _abi.ptr(hd), _abi.ptr(hw_),
                _abi.ptr(hm), H2, W2, _abi.ptr(vol), int(nhwc), b, k, h, w, D, stream), "dt_cv_mlp_hint_split_f32")
        elif planned:
            _abi.check(L.dt_cv_mlp_hint_planned_f32(
                _abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(pk["w1dyn"]), _abi.ptr(pk["w1pix"]),
                _abi.ptr(pk["w2p"]), _abi.ptr(pk["tail"]), _abi.ptr(hint_ptr), _abi.ptr(hd), _abi.ptr(hw_),
                _abi.ptr(hm), H2, W2, _abi.ptr(vol), int(nhwc), b, k, h, w, D, _abi.ptr(plan), int(plan.numel()), stream),
                "dt_cv_mlp_hint_planned_f32")
        elif _impl == "mfma":
            _abi.check(L.dt_cv_mlp_hint_f32(
                _abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(pk["w1dyn"]), _abi.ptr(pk["w1pix"]),
                _abi.ptr(pk["w2p"]), _abi.ptr(pk["tail"]), _abi.ptr(hint_ptr), _abi.ptr(hd), _abi.ptr(hw_),
                _abi.ptr(hm), H2, W2, _abi.ptr(vol), int(nhwc), b, k, h, w, D, stream), "dt_cv_mlp_hint_f32")
        elif _impl == "simple":
            r = pk["raw"]
            _abi.check(L.dt_cv_mlp_hint_simple_f32(
                _abi.ptr(cur), _abi.ptr(src_nhwc), _abi.ptr(params), _abi.ptr(self._planes_px), *[_abi.ptr(t) for t in r],
                _abi.ptr(hint_ptr), _abi.ptr(hd), _abi.ptr(hw_), _abi.ptr(hm), H2, W2, _abi.ptr(vol), b, k, c, h, w, D,
                stream), "dt_cv_mlp_hint_simple_f32")
        else:
            raise ValueError(_impl)
        if hook is not None or _graphs.recording() or "_stage_hook" in self.__dict__:
            _graphs.cut("mlp_end")
        if not _graphs.building():
            if hook is not None:
                hook("mlp_end")
            if gate is not None:
                gate("mlp_end")
        low = self._lowest(L, stream, vol, params, nhwc, dims)
        mask = None
        if return_mask:
            mask = self._mask(L, stream, params, per_view=not self._fast_mask, dims=dims)
        return vol, low, planes, mask

    def to_fast(self):
        """Reference feature_volume.py:358-365.  The fused kernel *is* the fast path; the returned
        manager shares the weights and differs only in the (reference-defined) mask semantics."""
        fast_cls = FastFeatureMeshHintVolumeManager if self._has_hint else FastFeatureVolumeManager
        m = fast_cls(self.matching_height, self.matching_width, num_depth_bins=self.num_depth_bins,
                     matching_dim_size=self.matching_dim_size, num_source_views=self.num_source_views)
        m.mlp = self.mlp
        if self._has_hint:
            m.hint_mlp = self.hint_mlp
        return m.to(next(self.mlp.parameters()).device)


class FastFeatureVolumeManager(FeatureVolumeManager):
    _fast_mask = True


class FeatureMeshHintVolumeManager(FeatureVolumeManager):
    """DoubleTake's volume: matching MLP + mesh-hint MLP (reference modules/mesh_hint_volume.py:12)."""

    _has_hint = True
    _fast_mask = False  # mesh_hint_volume.py:270-287: per-view mask of the last plane

    @torch.no_grad()
    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                cv_depth_hint_dict, depth_planes_bdhw=None, return_mask=False):
        """mesh_hint_volume.py:395-439 -> (cost_volume, lowest_cost, depth_planes_bdhw, overall_mask_bhw)."""
        return self._forward_impl(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth,
                                  max_depth, cv_depth_hint_dict, depth_planes_bdhw, return_mask)


class FastFeatureMeshHintVolumeManager(FeatureMeshHintVolumeManager):
    _fast_mask = True  # mesh_hint_volume.py:818-822
