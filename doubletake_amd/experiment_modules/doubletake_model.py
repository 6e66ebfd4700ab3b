"""Hot-path half of the reference's ``DepthModelCVHint`` (experiment_modules/doubletake_model.py).

The reference's forward (:265-425) is
    image encoder (timm) -> matching encoder -> cost volume -> CVEncoder -> decoder -> exp
The image encoder is a third-party timm network and stays outside this package (north star:
"PyTorch-ROCm for the ordinary 2D conv stacks").  ``DepthModelCVHint`` here owns the modules the
HIP kernels replace, under the reference's attribute names so a reference checkpoint's
``matching_model.* / cost_volume.* / cost_volume_net.* / depth_decoder.*`` keys load unchanged, and
exposes ``compute_matching_feats`` (:206-262) and ``forward_from_features`` = lines :341-349 and
:375-423 of the reference forward.  An image encoder can be attached as ``self.encoder`` (any
nn.Module producing the reference's five feature maps) to run the whole ``forward``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..modules import conv_ops as ops
from ..modules.cost_volume import CostVolumeManager, FeatureMeshHintVolumeManager, FeatureVolumeManager
from ..modules.networks import CVEncoder, DepthDecoderPP, ResnetMatchingEncoder
from ..modules.networks_fast import SkipDecoderRegression
from ..utils import graphs as _graphs
from ..utils.graphs import GraphedCallable
from ..utils.program import RecordedCallable

#: channel widths of the timm image encoders the reference uses (doubletake_model.py:121-130)
ENCODER_WIDTHS = {"resnet18d": [64, 64, 128, 256, 512], "efficientnet": [24, 48, 64, 160, 256]}


class MatchingFeatureCache:
    """LRU map frame id -> matching feature map [C,h,w] (device tensor).  At 640x480 one entry is 1.2 MB, so
    the default capacity (8192 keyframes, 10 GB) holds any ScanNet scan many times over in 288 GB of HBM."""

    def __init__(self, capacity=8192):
        from collections import OrderedDict

        self.capacity = capacity
        self.token = None  # (weights version, image shape, device) the entries were computed with
        self._store = OrderedDict()
        self.hits = 0
        self.misses = 0

    def __contains__(self, fid):
        return fid in self._store

    def __len__(self):
        return len(self._store)

    def get(self, fid):
        """Entry for frame ``fid``; if it was produced on another stream (``prefetch_matching_feats``) the current
        stream is made to wait for it first."""
        self._store.move_to_end(fid)
        self.hits += 1
        feat, ready = self._store[fid]
        if ready is not None:
            from .. import _abi

            _abi.wait_ready(ready, feat.device)
            # the entry lives in the producer stream's allocator pool; a consumer on ANOTHER stream (several keyframes in
            # flight) must be known to the allocator whether or not the producer has finished: an eviction, a put() of the
            # same frame id or clear() while this stream's kernels are still queued would otherwise hand the block back to
            # the producer's pool under their reads
            cur = torch.cuda.current_stream(feat.device)
            if cur.cuda_stream != ready[1]:
                feat.record_stream(cur)
        return feat

    def put(self, fid, feat, ready=None):
        self.misses += 1
        self._store[fid] = (feat, ready)
        self._store.move_to_end(fid)
        while len(self._store) > self.capacity:
            self._store.popitem(last=False)

    def clear(self):
        self._store.clear()


#: feature_volume_type -> manager class.  DepthModel (SimpleRecon, sr_depth_model.py:186-194) accepts the first two,
#: DepthModelCVHint (doubletake_model.py:172-177) the third.
VOLUME_CLASSES = {
    "simple_cost_volume": CostVolumeManager,
    "mlp_feature_volume": FeatureVolumeManager,
    "mlp_mesh_hint_feature_volume": FeatureMeshHintVolumeManager,
}


class _HotPathDepthModel(nn.Module):
    """What DepthModel (experiment_modules/sr_depth_model.py) and DepthModelCVHint (doubletake_model.py) share: the
    modules the HIP kernels replace under the reference's attribute names, compute_matching_feats, and the forward
    from encoder features on.  Subclasses state which ``feature_volume_type`` values they accept and the default."""

    #: option fields the reference constructors read (doubletake_model.py:84-204, sr_depth_model.py:100-230)
    _OPT_FIELDS = ("image_height", "image_width", "image_encoder_name", "depth_decoder_name", "matching_num_depth_bins",
                   "matching_scale", "matching_feature_dims", "model_num_views", "min_matching_depth", "max_matching_depth",
                   "matching_encoder_type", "feature_volume_type")
    _VOLUME_TYPES = ()
    _DEFAULT_VOLUME_TYPE = None

    def __init__(self, image_height=384, image_width=512, image_encoder_name="resnet18d", depth_decoder_name="skip",
                 matching_num_depth_bins=64, matching_scale=1, matching_feature_dims=16, model_num_views=8,
                 min_matching_depth=0.25, max_matching_depth=5.0, matching_encoder_type="resnet", feature_volume_type=None):
        """Keyword form, or the reference's ``Model(opts)`` with an options object / namespace that carries the fields
        of ``_OPT_FIELDS`` (missing ones keep the reference defaults of options.py)."""
        super().__init__()
        if not isinstance(image_height, int) and hasattr(image_height, "image_height"):
            opts = image_height
            for bad, want in (("cv_encoder_type", "multi_scale_encoder"), ("loss_type", "log_l1")):
                if getattr(opts, bad, want) != want:
                    raise ValueError(f"Unrecognized option {getattr(opts, bad)!r} for {bad} (the reference builds only {want!r})")
            self.run_opts = opts
            image_height = opts.image_height
            image_width = getattr(opts, "image_width", image_width)
            image_encoder_name = getattr(opts, "image_encoder_name", image_encoder_name)
            depth_decoder_name = getattr(opts, "depth_decoder_name", depth_decoder_name)
            matching_num_depth_bins = getattr(opts, "matching_num_depth_bins", matching_num_depth_bins)
            matching_scale = getattr(opts, "matching_scale", matching_scale)
            matching_feature_dims = getattr(opts, "matching_feature_dims", matching_feature_dims)
            model_num_views = getattr(opts, "model_num_views", model_num_views)
            min_matching_depth = getattr(opts, "min_matching_depth", min_matching_depth)
            max_matching_depth = getattr(opts, "max_matching_depth", max_matching_depth)
            matching_encoder_type = getattr(opts, "matching_encoder_type", matching_encoder_type)
            feature_volume_type = getattr(opts, "feature_volume_type", feature_volume_type)
        if feature_volume_type is None:
            feature_volume_type = self._DEFAULT_VOLUME_TYPE
        if feature_volume_type not in self._VOLUME_TYPES:
            # same refusal as the reference (sr_depth_model.py:190-194, doubletake_model.py:174-177)
            raise ValueError(f"Unrecognized option {feature_volume_type} for feature volume type! "
                             f"{type(self).__name__} builds {', '.join(self._VOLUME_TYPES)}")
        self.feature_volume_type = feature_volume_type
        key = "efficientnet" if "efficientnet" in image_encoder_name else "resnet18d"
        self.num_ch_enc = list(ENCODER_WIDTHS[key])
        self.matching_scale = matching_scale
        self.min_matching_depth = min_matching_depth
        self.max_matching_depth = max_matching_depth
        self.cost_volume_net = CVEncoder(
            num_ch_cv=matching_num_depth_bins, num_ch_enc=self.num_ch_enc[matching_scale:], num_ch_outs=[64, 128, 256, 384])
        dec_in = self.num_ch_enc[:matching_scale] + self.cost_volume_net.num_ch_enc
        if depth_decoder_name == "unet_pp":
            self.depth_decoder = DepthDecoderPP(dec_in)
        elif depth_decoder_name == "skip":
            self.depth_decoder = SkipDecoderRegression(dec_in)
        else:
            raise ValueError("Unrecognized option for depth decoder name!")
        self.cost_volume = VOLUME_CLASSES[feature_volume_type](
            matching_height=image_height // (2 ** (matching_scale + 1)),
            matching_width=image_width // (2 ** (matching_scale + 1)),
            num_depth_bins=matching_num_depth_bins, matching_dim_size=matching_feature_dims,
            num_source_views=model_num_views - 1)
        self.encoder = None
        self.matching_feature_cache = MatchingFeatureCache()
        if matching_encoder_type == "resnet":  # doubletake_model.py:196-197
            self.matching_model = ResnetMatchingEncoder(18, matching_feature_dims, pretrained=False)
        elif matching_encoder_type is None:
            self.matching_model = None
        else:
            raise ValueError(f"Unrecognized option {matching_encoder_type} for matching encoder type!")

    @torch.no_grad()
    def compute_matching_feats(self, cur_image, src_image, unbatched_matching_encoder_forward=False, cur_ids=None,
                               src_ids=None, scan_ids=None):
        """doubletake_model.py:206-262: matching features of the current image [b,3,H,W] and the source
        images [b,K,3,H,W] -> ([b,C,h,w], [b,K,C,h,w]).  Batched: all b*(1+K) images in one pass.

        Extension ("cross-frame feature caching"): with ``cur_ids`` (b frame-id strings)
        and ``src_ids`` (K lists of b strings, the layout of ``src_data["frame_id_string"]``) features are
        kept in an HBM-resident LRU cache and only frames not seen before go through the encoder -- in a
        scan every keyframe is encoded once instead of once per tuple it appears in.  Frame ids repeat across
        scans ("000012" exists in every ScanNet scan), so entries are keyed by (scan id, frame id): pass
        ``scan_ids`` (one string, or b strings) whenever one model instance serves more than one scan.  The cache
        empties itself when the matching encoder's weights change (load_state_dict, in-place update, .to()).
        Features of a cached
        frame were computed in a different batch, so they can differ from the batched result in the last
        bits (different K-split of the same fp32 sums), exactly like the reference's unbatched flag."""
        if self.matching_model is None:
            raise RuntimeError("this model was built without a matching encoder")
        b, m = cur_image.shape[0], src_image.shape[1] + 1
        if cur_ids is not None and src_ids is not None:
            ids = [[cur_ids[i]] + [src_ids[k][i] for k in range(m - 1)] for i in range(b)]
            scans = [scan_ids] * b if isinstance(scan_ids, str) or scan_ids is None else list(scan_ids)
            flat_ids = [(scans[i], fid) for i, row in enumerate(ids) for fid in row]
            cache = self.matching_feature_cache
            # entries are only valid for the weights (and the image size) they were computed with
            token = self._cache_token(cur_image.shape[1:], cur_image.device)
            if cache.token != token:
                cache.clear()
                cache.token = token
            # (entries this call needs are taken out of the cache BEFORE anything new goes in: with a capacity below
            #  b * (1 + K) the insertions below could otherwise evict them)
            have = {fid: cache.get(fid) for fid in dict.fromkeys(flat_ids) if fid in cache}
            missing = [j for j, fid in enumerate(flat_ids) if fid not in have]
            first = {}
            for j in missing:  # the same new frame may appear twice in one batch
                first.setdefault(flat_ids[j], j)
            todo = sorted(first.values())
            if todo:
                # only the images that are actually encoded are gathered (views of the two input tensors: no copy of the
                # whole [b, 1+K] image tuple, which is 29 MB per keyframe at 640x480)
                pick = lambda j: cur_image[j // m] if j % m == 0 else src_image[j // m, j % m - 1]
                imgs = pick(todo[0]).unsqueeze(0) if len(todo) == 1 else torch.stack([pick(j) for j in todo], 0)
                new = self._encode(imgs)
                from .. import _abi

                # (ready token: a later keyframe processed on ANOTHER stream -- several keyframes in flight -- orders itself
                #  behind this encoder pass when it takes the entry; free on the producing stream)
                ready = _abi.record_ready(new.device)
                for row, j in enumerate(todo):
                    have[flat_ids[j]] = new[row]
                    cache.put(flat_ids[j], new[row], ready)
            # the cached maps are gathered straight into the two tensors the volume takes (one stack for the source views,
            # none for the current view at batch 1) -- not stacked as one [b*(1+K)] tensor and sliced apart again
            got = [have[fid] for fid in flat_ids]
            cur_f = got[0].unsqueeze(0) if b == 1 else torch.stack([got[i * m] for i in range(b)], 0)
            src_f = torch.stack([got[i * m + 1 + k] for i in range(b) for k in range(m - 1)], 0)
            return cur_f, src_f.view(b, m - 1, *src_f.shape[1:])
        else:
            flat = torch.cat([cur_image.unsqueeze(1), src_image], dim=1).flatten(0, 1)
            if unbatched_matching_encoder_forward:
                feats = torch.cat([self.matching_model(f) for f in flat.split(1, dim=0)], dim=0)
            else:
                feats = self.matching_model(flat)
        feats = feats.view(b, m, *feats.shape[1:])
        return feats[:, 0], feats[:, 1:].contiguous()

    def _cache_token(self, image_shape_chw, device):
        return (self._param_token("_mt_cache", ("matching_model",)), tuple(image_shape_chw), str(device))

    @torch.no_grad()
    def prefetch_matching_feats(self, image_n3hw, frame_ids, scan_ids=None, stream=None):
        """Encode keyframes AHEAD of the frame that needs them, on a side stream, into the feature cache.

        In the incremental mode frame t cannot start before frame t-1's TSDF update (hint), but its matching features
        depend on the image alone: encoded on a second HIP stream while frame t-1's latency-bound conv stack leaves most
        of the chip idle, the 0.40 ms of the matching encoder disappear from the per-frame critical path
        (``loops.run_incremental_scan(..., lookahead=...)``).  ``compute_matching_feats(..., cur_ids=, src_ids=)`` finds
        the entries and orders its stream behind the encoder through the entry's event.  Frames already cached are skipped.
        Returns the number of frames encoded."""
        if self.matching_model is None:
            raise RuntimeError("this model was built without a matching encoder")
        from .. import _abi

        dev = image_n3hw.device
        n = image_n3hw.shape[0]
        scans = [scan_ids] * n if isinstance(scan_ids, str) or scan_ids is None else list(scan_ids)
        keys = [(scans[i], frame_ids[i]) for i in range(n)]
        cache = self.matching_feature_cache
        token = self._cache_token(image_n3hw.shape[1:], dev)
        if cache.token != token:
            cache.clear()
            cache.token = token
        first = {}
        for j, key in enumerate(keys):
            if key not in cache:
                first.setdefault(key, j)
        todo = sorted(first.values())
        if not todo:
            return 0
        if stream is None:
            stream = getattr(self, "_lookahead_stream", None)
            if stream is None or stream.device != dev:
                stream = self._lookahead_stream = torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        stream.wait_stream(cur)  # the images (and the weights) are ready on the caller's stream
        with torch.cuda.stream(stream):
            # (slices / the tensor itself where possible: indexing with a Python list uploads an index tensor from
            #  pageable host memory, which costs the host the better part of a millisecond per frame)
            if len(todo) == 1:
                imgs = image_n3hw[todo[0]:todo[0] + 1]
            elif todo == list(range(n)):
                imgs = image_n3hw
            else:
                imgs = torch.stack([image_n3hw[j] for j in todo], 0)
            new = self._encode(imgs)
            ready = _abi.record_ready(dev)
        image_n3hw.record_stream(stream)
        for row, j in enumerate(todo):
            cache.put(keys[j], new[row], ready)
        return len(todo)

    # ---- hipGraph replay (opt-in) ----------------------------------------------------------------------------------
    _HINT_KEYS = ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw")

    def enable_hip_graphs(self, on=True):
        """Replay ``forward_from_features`` (and the single-image matching-encoder pass of the feature cache) from captured
        hipGraphs instead of launching ~50 kernels from Python per keyframe (utils/graphs.py).  For fixed-shape loops --
        above all the incremental mode, whose per-frame host time otherwise exceeds the GPU time.  The returned tensors
        are then STATIC buffers, overwritten by the next call: consume or clone them first (the per-scan loops of
        ``doubletake_amd.loops`` do).  Weights may change between calls (the graphs are keyed on their versions)."""
        def between(tag):
            if tag == "after_volume":
                hook = self.__dict__.pop("after_volume", None)
                if hook is not None:
                    hook()
            else:  # "mlp_begin" / "mlp_end": the event hook bench.py brackets the dominant kernel with, and the pipeline's gate
                self._volume_hooks(tag)

        def cut_config():
            # which cuts the eager function would place right now: part of the graph cache key, so that a graph captured
            # before a hook was installed is not replayed (its hook never called) once one is; and the process-wide
            # switches that decide which kernels the captured launches are
            from ..modules.cost_volume import FeatureVolumeManager

            return ("after_volume" in self.__dict__, FeatureVolumeManager._event_hook is not None or
                    "_stage_hook" in self.cost_volume.__dict__, self._launch_config())

        self._graphed_forward = (GraphedCallable(self._forward_from_features_eager, between=between, cut_config=cut_config)
                                 if on else None)
        self._graphed_encoder = GraphedCallable(lambda img: self.matching_model(img)) if on and self.matching_model is not None else None
        return self

    def _volume_hooks(self, tag):
        """What the eager volume manager calls around its kernel (cost_volume._forward_impl), for the replay mechanisms'
        ``between`` callbacks: the pipeline's gate outside, the timing hook inside."""
        from ..modules.cost_volume import FeatureVolumeManager

        gate = self.cost_volume.__dict__.get("_stage_hook")
        hook = FeatureVolumeManager._event_hook
        first, second = (gate, hook) if tag == "mlp_begin" else (hook, gate)
        if first is not None:
            first(tag)
        if second is not None:
            second(tag)

    def _launch_config(self):
        cv = self.cost_volume
        return ops.launch_config() + (getattr(cv, "precision", None), getattr(cv, "use_span_plan", None),
                                      getattr(cv, "channels_last_output", None))

    def enable_launch_programs(self, on=True):
        """Replay ``forward_from_features`` from a launch program recorded at the C ABI (utils/program.py, csrc/program.hip):
        the ~50 kernel launches of cost volume + CVEncoder + decoder + heads are recorded once per (input signature, stream) and
        re-issued by ``dt_program_launch`` -- one host call per segment instead of one per kernel plus the module code
        around it (0.8 ms -> 0.2 ms of host time per keyframe at 640x480).  Same kernels, same arguments: bit-identical to the
        eager path.  The returned tensors are STATIC buffers of the (signature, stream) pair, overwritten by the next call
        with that pair: consume or clone them first (``parallel.KeyframePipeline`` and the per-scan loops do).  The program is
        cut around the volume kernel and behind the volume stage, so the event hook (bench.py) and the one-shot
        ``after_volume`` hook (loops.matching_lookahead) work without a new recording.  Weights may change between calls
        (programs are keyed on their versions); inputs must be dense fp32 (NHWC pyramids) -- what the eager path accepts
        without converting."""
        def between(tag):
            if tag == "after_volume":
                hook = self.__dict__.pop("after_volume", None)
                if hook is not None:
                    hook()
            else:
                self._volume_hooks(tag)

        if on and getattr(self, "_recorded_forward", None) is not None:
            return self  # (already on: keep the recorded programs)
        self._recorded_forward = (RecordedCallable(self._forward_from_features_eager, between=between,
                                                   cut_config=self._launch_config) if on else None)
        # the single-image matching-encoder pass of the feature cache (the incremental loop encodes exactly one new keyframe
        # per frame: 14 launches)
        self._recorded_encoder = (RecordedCallable(lambda img: self.matching_model(img), cut_config=ops.launch_config)
                                  if on and self.matching_model is not None else None)
        return self

    def _weights_token(self):
        """(data_ptr, version) of every parameter the replayed launches depend on.  data_ptr as well as the version counter:
        ``module.weight = nn.Parameter(...)`` or a swapped submodule brings a new tensor at an old version, a device move a new
        address, and the captured launches carry the old packed-weight pointers.
        The module trees are not walked on every call (round 6: 0.15 ms of the 0.34 ms the host needs per step once the model
        step is one C call): the walk is cached as the lists of (``_modules`` dict, name, child) and (``_parameters`` dict,
        name, parameter) edges it followed, and a call first checks that every edge still holds the same object -- a parameter
        or submodule assigned since then fails the check and triggers a new walk."""
        return self._param_token("_wt_cache", ("cost_volume", "cost_volume_net", "depth_decoder"))

    def _param_token(self, slot, root_names):
        """(data_ptr, version) of every parameter below the named child modules, from a cached walk (see _weights_token)."""
        c = self.__dict__.get(slot)
        if c is not None:
            for d, k, obj in c[0]:
                if d.get(k) is not obj:
                    c = None
                    break
        if c is None:
            edges, params = [], []
            stack = []
            for n in root_names:
                m = self._modules.get(n)
                edges.append((self._modules, n, m))
                if m is not None:
                    stack.append(m)
            seen = set()
            while stack:
                m = stack.pop()
                if id(m) in seen:
                    continue
                seen.add(id(m))
                for n, p in m._parameters.items():
                    edges.append((m._parameters, n, p))
                    if p is not None:
                        params.append(p)
                for n, ch in m._modules.items():
                    edges.append((m._modules, n, ch))
                    if ch is not None:
                        stack.append(ch)
            c = self.__dict__[slot] = (edges, params)
        return tuple((p.data_ptr(), p._version) for p in c[1])

    def _encode(self, images_n3hw):
        """Matching encoder pass; single images replay a captured graph when graphs are on (the incremental loop encodes
        exactly one new keyframe per frame)."""
        g = getattr(self, "_recorded_encoder", None)
        if g is None:
            g = getattr(self, "_graphed_encoder", None)
        if g is not None and images_n3hw.shape[0] == 1:
            token = self._param_token("_mt_cache", ("matching_model",))
            if getattr(self, "_encoder_token", token) != token:
                g.reset()
            self._encoder_token = token
            return g(images_n3hw).clone()  # (the cache keeps the features: they must not alias the graph's static output)
        return self.matching_model(images_n3hw)

    @torch.no_grad()
    def forward_from_features(self, cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam,
                              cur_cam_T_src_cam, src_K, cur_invK, cv_depth_hint_dict=None, return_mask=False):
        """cur_feats: list of 5 image-prior maps (strides 2..32); matching feats at stride 4.
        Returns the reference's output dict (doubletake_model.py:410-423)."""
        g = getattr(self, "_recorded_forward", None)
        if g is None:
            g = getattr(self, "_graphed_forward", None)
        if g is None:
            return self._forward_from_features_eager(cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam,
                                                     cur_cam_T_src_cam, src_K, cur_invK, cv_depth_hint_dict, return_mask)
        token = self._weights_token()
        if getattr(self, "_forward_token", token) != token:
            g.reset()  # packed weights are baked into the captured launches
        self._forward_token = token
        hint = None
        if cv_depth_hint_dict is not None and isinstance(self.cost_volume, FeatureMeshHintVolumeManager):
            hint = {k: cv_depth_hint_dict[k] for k in self._HINT_KEYS}  # (the drivers pass the whole cur_data dict)
        return dict(g(list(cur_feats), matching_cur_feats, matching_src_feats, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K,
                      cur_invK, hint, bool(return_mask)))

    @torch.no_grad()
    def _forward_from_features_eager(self, cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam,
                                     cur_cam_T_src_cam, src_K, cur_invK, cv_depth_hint_dict=None, return_mask=False):
        dev = matching_cur_feats.device
        key = (str(dev), self.min_matching_depth, self.max_matching_depth)
        if getattr(self, "_depth_range_key", None) != key:  # device-resident constants, built once
            self._min_depth = torch.tensor(self.min_matching_depth, device=dev, dtype=torch.float32).view(1, 1, 1, 1)
            self._max_depth = torch.tensor(self.max_matching_depth, device=dev, dtype=torch.float32).view(1, 1, 1, 1)
            self._depth_range_key = key
        min_depth, max_depth = self._min_depth, self._max_depth
        kw = dict(cur_feats=matching_cur_feats, src_feats=matching_src_feats, src_extrinsics=src_cam_T_cur_cam,
                  src_poses=cur_cam_T_src_cam, src_Ks=src_K, cur_invK=cur_invK, min_depth=min_depth, max_depth=max_depth,
                  return_mask=return_mask)
        if isinstance(self.cost_volume, FeatureMeshHintVolumeManager):  # the other managers take no hints
            kw["cv_depth_hint_dict"] = cv_depth_hint_dict
        cost_volume, lowest_cost, overall_mask = self.volume_stage(kw)
        # graph mode: the replay is split here so that the hook below runs between the halves -- only when a hook is
        # waiting at capture time (every cut is one more hipGraphLaunch per replay)
        if "after_volume" in self.__dict__ or _graphs.recording():
            _graphs.cut("after_volume")  # (launch programs: always -- a segment is one more C call, not a hipGraphLaunch)
        hook = None if _graphs.building() else self.__dict__.pop("after_volume", None)
        if hook is not None:
            # one-shot: work for the caller to enqueue on ANOTHER stream behind the volume kernel (which fills every CU
            # and all of its LDS) and beside the conv stack that follows (latency-bound, most of the chip idle):
            # loops.matching_lookahead encodes the next frame's keyframe here
            hook()
        return self.network_stage(cur_feats, cost_volume, lowest_cost, overall_mask)

    # The two halves of the forward pass, callable on their own so that a driver with several keyframes in flight can order
    # them across HIP streams (bench.py --schedule phased): the volume kernel owns every CU and all of its LDS, the conv
    # stack that follows is a chain of latency-bound launches that several frames can share.
    @torch.no_grad()
    def volume_stage(self, volume_kwargs):
        """cost volume + lowest cost + mask of one keyframe batch (kwargs of self.cost_volume)."""
        cost_volume, lowest_cost, _, overall_mask = self.cost_volume(**volume_kwargs)
        return cost_volume, lowest_cost, overall_mask

    @torch.no_grad()
    def network_stage(self, cur_feats, cost_volume, lowest_cost, overall_mask):
        """CVEncoder + depth decoder + exp on the volume of volume_stage -> the reference's output dict."""
        cv_feats = self.cost_volume_net(cost_volume, cur_feats[self.matching_scale:])
        feats = list(cur_feats[: self.matching_scale]) + cv_feats
        depth_outputs = self.depth_decoder(feats, with_depth=True)  # heads write exp(log depth) themselves
        for k in list(depth_outputs.keys()):
            if not k.startswith("log_depth"):
                continue
            log_depth = depth_outputs[k].float()
            depth_outputs[k] = log_depth
            if k.replace("log_", "") not in depth_outputs:
                depth_outputs[k.replace("log_", "")] = ops.exp(log_depth)
        depth_outputs["lowest_cost_bhw"] = lowest_cost
        depth_outputs["overall_mask_bhw"] = overall_mask
        return depth_outputs

    @staticmethod
    @torch.no_grad()
    def relative_poses(cur_data, src_data):
        """(src_cam_T_cur_cam, cur_cam_T_src_cam), reference doubletake_model.py:330-339 -- there two torch.matmul calls on
        [b,K,4,4] tensors; one HIP launch here (a 4x4 matmul through hipBLASLt costs ~100 us of host time on ROCm)."""
        from .. import _abi

        src_cTw, src_wTc = src_data["cam_T_world_b44"], src_data["world_T_cam_b44"]
        cur_cTw, cur_wTc = cur_data["cam_T_world_b44"], cur_data["world_T_cam_b44"]
        if not src_cTw.is_cuda:
            raise _abi.DoubletakeHipError("camera matrices are on the CPU; doubletake_amd only runs on a ROCm GPU (no CPU fallback)")
        f = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        src_cTw, src_wTc, cur_cTw, cur_wTc = f(src_cTw), f(src_wTc), f(cur_cTw), f(cur_wTc)
        b, k = src_cTw.shape[:2]
        ext = torch.empty_like(src_cTw)
        poses = torch.empty_like(src_cTw)
        _abi.check(_abi.lib().dt_cv_relative_poses_f32(_abi.ptr(src_cTw), _abi.ptr(src_wTc), _abi.ptr(cur_cTw), _abi.ptr(cur_wTc),
                                                       b, k, _abi.ptr(ext), _abi.ptr(poses), _abi.current_stream(src_cTw.device)),
                   "dt_cv_relative_poses_f32")
        return ext, poses

    @torch.no_grad()
    def forward(self, phase, cur_data, src_data, unbatched_matching_encoder_forward=False, return_mask=False):
        """Reference signature (doubletake_model.py:265).  Needs self.encoder (the timm image encoder)."""
        if self.encoder is None:
            raise RuntimeError("attach .encoder (the image-prior network, a PyTorch-ROCm module) or call forward_from_features")
        s = self.matching_scale
        src_K = src_data[f"K_s{s}_b44"]
        cur_invK = cur_data[f"invK_s{s}_b44"]
        src_cam_T_cur_cam, cur_cam_T_src_cam = self.relative_poses(cur_data, src_data)
        cur_feats = self.encoder(cur_data["image_b3hw"])
        ids = {}
        if getattr(self, "use_feature_cache", False) and "frame_id_string" in cur_data and "frame_id_string" in src_data:
            # opt-in (model.use_feature_cache = True): keyframes are encoded once per scan and kept in HBM; the drivers'
            # batches carry the ids (datasets ... pass_frame_id=True, test_incremental.py:150)
            ids = dict(cur_ids=list(cur_data["frame_id_string"]), src_ids=[list(v) for v in src_data["frame_id_string"]],
                       scan_ids=cur_data.get("scan_id_string"))
        m_cur, m_src = self.compute_matching_feats(cur_data["image_b3hw"], src_data["image_b3hw"],
                                                   unbatched_matching_encoder_forward, **ids)
        return self.forward_from_features(cur_feats, m_cur, m_src, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K,
                                          cur_invK, cur_data, return_mask=return_mask)


class DepthModelCVHint(_HotPathDepthModel):
    """DoubleTake (experiment_modules/doubletake_model.py): the mesh-hint feature volume; ``cur_data`` carries the hint
    maps (depth_hint_b1hw, sampled_weights_b1hw, depth_hint_mask_b1hw)."""

    _VOLUME_TYPES = ("mlp_mesh_hint_feature_volume",)
    _DEFAULT_VOLUME_TYPE = "mlp_mesh_hint_feature_volume"


class DepthModel(_HotPathDepthModel):
    """SimpleRecon (experiment_modules/sr_depth_model.py:186-204), ``model_type == "depth_model"``: the dot-product
    ``simple_cost_volume`` or the metadata ``mlp_feature_volume`` (the options.py default), no hints.  Same modules,
    state-dict keys and forward contract (:283-420) as the reference class."""

    _VOLUME_TYPES = ("simple_cost_volume", "mlp_feature_volume")
    _DEFAULT_VOLUME_TYPE = "mlp_feature_volume"
