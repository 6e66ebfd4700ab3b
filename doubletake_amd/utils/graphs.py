"""hipGraph replay of a fixed-shape call: capture once, replay with one host call.

The hot path at batch 1 is a chain of ~50 short kernels per keyframe; launched eagerly from Python the HOST needs about
1 ms per frame for them (ctypes call + tensor bookkeeping per launch; 1.4 ms for a whole frame of the incremental loop).  A
captured graph is replayed with a single hipGraphLaunch (0.5 ms of host time per incremental frame).  Opt-in: on an idle host
the eager loop is GPU-bound anyway and measured slightly FASTER than the replay (2.03 vs 2.12 ms per incremental frame,
1.76 vs 1.79 ms per bench step: DESIGN.md 4.2, round 3) -- the replay is for deployments whose host is busy or slow.

``GraphedCallable(fn)`` wraps a function of tensors (positional / keyword arguments, nested in lists, tuples and dicts;
non-tensor arguments are part of the signature).  For every distinct signature (shapes, dtypes, devices, constants) it
  1. copies the arguments into static buffers it owns,
  2. runs ``fn`` twice on a side stream (weight packs, per-stream library scratch, allocator pools),
  3. captures ``fn`` into a graph (``torch.cuda.graph``: hipStreamBeginCapture on ROCm),
and afterwards copies the new arguments into the static buffers (skipped when the caller hands over the static buffer
itself) and replays.  The outputs are the SAME tensors on every call: consume (or clone) them before the next call with that
signature.  One replay at a time per signature (the graph owns its intermediates and the library scratch of its capture
stream).  No CPU fallback: capture needs a ROCm GPU."""
from __future__ import annotations

import threading

import torch


def _flatten(obj, out):
    """Tensors of a nested structure in a fixed order; returns a hashable skeleton describing everything else."""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
        return ("T", tuple(obj.shape), str(obj.dtype), str(obj.device), tuple(obj.stride()))
    if isinstance(obj, (list, tuple)):
        return (type(obj).__name__,) + tuple(_flatten(o, out) for o in obj)
    if isinstance(obj, dict):
        return ("dict",) + tuple((k, _flatten(obj[k], out)) for k in sorted(obj, key=str))
    return ("C", obj if isinstance(obj, (int, float, str, bool, type(None))) else id(obj))


def _rebuild(obj, tensors):
    """The same structure with its tensors replaced, in _flatten order, by the next items of the iterator ``tensors``."""
    if isinstance(obj, torch.Tensor):
        return next(tensors)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_rebuild(o, tensors) for o in obj)
    if isinstance(obj, dict):
        rebuilt = {k: _rebuild(obj[k], tensors) for k in sorted(obj, key=str)}
        return {k: rebuilt[k] for k in obj}
    return obj


_TLS = threading.local()  # .capturing: the GraphedCallable / RecordedCallable whose capture (recording) is in progress on THIS
#                            thread (cut() talks to it); .warming: True while its un-captured warm-up passes run; .recording:
#                            True while a RecordedCallable records (launches execute AND are recorded: no stream capture)


def cut(tag=None):
    """Called by the wrapped function at a point where the replay should be split in two graphs, so that the owner can
    enqueue other work between them -- an event, or a launch on another stream: ``GraphedCallable(fn, between=...)``
    receives ``tag`` after the segment that ends here.  A no-op outside a capture.  Every cut costs one more
    hipGraphLaunch per replay: callers cut only where something is actually going to be enqueued."""
    g = getattr(_TLS, "capturing", None)
    if g is not None:
        g._cut(tag)


def recording():
    """True while a RecordedCallable (utils/program.py) records the wrapped function's launches on this thread."""
    return bool(getattr(_TLS, "recording", False))


def building():
    """True while the wrapped function runs for a replay mechanism rather than for its caller: hipGraph capture, launch-program
    recording, or the warm-up passes in front of either.  Hooks that the replay's ``between`` callback will call (events,
    one-shot work for another stream) must not fire then."""
    return torch.cuda.is_current_stream_capturing() or recording() or in_warmup()


def in_warmup():
    """True inside the warm-up passes a GraphedCallable runs before its capture (side stream, throw-away outputs): one-shot
    hooks of the wrapped function must not be consumed there."""
    return bool(getattr(_TLS, "warming", False))


class GraphedCallable:
    def __init__(self, fn, warmup=2, between=None, cut_config=None):
        """between(tag): called during replay after every graph segment that a ``cut(tag)`` call in fn ended.
        cut_config(): hashable describing which cuts fn WOULD make right now (e.g. which hooks are installed); it is part of
        the cache key, so a graph captured without a cut is never replayed for a call that needs one (and vice versa)."""
        self.fn = fn
        self.warmup = max(2, int(warmup))
        self.between = between
        self.cut_config = cut_config
        self._entries = {}
        self.captures = 0
        self.replays = 0
        self._segments = None
        self._tags = None
        self._pool = None

    def _cut(self, tag):
        self._tags.append(tag)
        self._segments[-1].capture_end()
        g = torch.cuda.CUDAGraph()
        self._segments.append(g)
        g.capture_begin(pool=self._pool)

    def _capture(self, args, kwargs, tensors):
        dev = tensors[0].device
        static = [t.detach().clone(memory_format=torch.preserve_format) for t in tensors]
        s_args, s_kwargs = _rebuild((args, kwargs), iter(static))
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        _TLS.warming = True
        try:
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self.fn(*s_args, **s_kwargs)
                    # the second pass finds every weight-pack cache entry complete (their ready events have fired), so
                    # nothing in the captured pass has to look at an event recorded outside the capture
                    side.synchronize()
        finally:
            _TLS.warming = False
        torch.cuda.current_stream(dev).wait_stream(side)
        # capture ON the warm-up stream: the library keeps its cross-workgroup reduction scratch per (device, stream) and
        # must not allocate while a stream is capturing
        self._pool = torch.cuda.graph_pool_handle()
        self._segments = [torch.cuda.CUDAGraph()]
        self._tags = []
        side.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        _TLS.capturing = self
        try:
            with torch.cuda.stream(side):
                self._segments[0].capture_begin(pool=self._pool)
                try:
                    out = self.fn(*s_args, **s_kwargs)
                finally:
                    self._segments[-1].capture_end()
        finally:
            _TLS.capturing = None
        torch.cuda.current_stream(dev).wait_stream(side)
        segments, self._segments = self._segments, None
        self.captures += 1
        return dict(graphs=segments, tags=list(self._tags), static=static, out=out)

    def __call__(self, *args, **kwargs):
        tensors = []
        sig = _flatten((args, kwargs), tensors)
        if not tensors or not tensors[0].is_cuda:
            raise RuntimeError("GraphedCallable needs ROCm GPU tensors (no CPU fallback)")
        if self.cut_config is not None:
            sig = (sig, self.cut_config())
        ent = self._entries.get(sig)
        if ent is None:
            ent = self._entries[sig] = self._capture(args, kwargs, tensors)
        for dst, src in zip(ent["static"], tensors):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        tags = ent["tags"]
        for i, g in enumerate(ent["graphs"]):
            g.replay()
            if i < len(tags) and self.between is not None:
                self.between(tags[i])
        self.replays += 1
        return ent["out"]

    def static_inputs(self, *args, **kwargs):
        """The static input buffers (same nesting as the arguments) of the signature these arguments have -- capturing it
        first if needed.  A producer that writes straight into them saves the per-call copies."""
        tensors = []
        sig = _flatten((args, kwargs), tensors)
        if self.cut_config is not None:
            sig = (sig, self.cut_config())
        ent = self._entries.get(sig)
        if ent is None:
            ent = self._entries[sig] = self._capture(args, kwargs, tensors)
        return _rebuild((args, kwargs), iter(ent["static"]))

    def reset(self):
        self._entries.clear()
