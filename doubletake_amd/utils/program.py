"""Launch programs: a fixed-shape call recorded once at the C ABI, replayed with ONE host call (csrc/program.hip).

The hot path at batch 1 is a chain of ~50 short kernels per keyframe.  Enqueued eagerly the host needs 0.8 ms per keyframe
for them (module code, plan look-ups, one ctypes call and a few tensor allocations per launch) against 1.3 ms of GPU time;
a hipGraph replay (utils/graphs.py) has the single call but measured slower on the wall than the eager loop on this runtime.
``RecordedCallable`` keeps the graph's contract and replaces the mechanism: the wrapped function runs once while the
library records every kernel launch it makes on the current stream (``dt_program_begin`` .. ``dt_program_end``: kernel
address, grid, argument bytes), and every later call with the same signature is ``dt_program_launch`` -- the recorded
``hipLaunchKernel`` calls re-issued from C with the new input addresses patched in.

Contract (the same as GraphedCallable's, plus two points):
  * ``fn`` takes tensors (positional / keyword, nested in lists, tuples, dicts; everything else is part of the signature)
    and returns tensors (nested likewise).  The returned tensors are the SAME buffers on every call with that signature on
    that stream: consume (or clone) them before the next such call.
  * a program belongs to the stream it was recorded on (the library's split-K scratch is per stream): calling from another
    stream records another program.  N keyframes in flight = N streams = N programs with their own buffers.
  * the inputs are NOT copied: their addresses are patched into the recorded arguments, so shapes, strides and dtypes are
    part of the signature and the tensors must not alias each other.
  * ``fn`` may only launch through the library.  A torch op that launches a kernel of its own (a cast, a copy, an
    interpolate) would run while recording and be missing from every replay: the recording pass runs under a dispatch
    guard that raises ``NotReplayable`` naming the op instead.  Views and allocations are fine.
Intermediates and outputs are allocated from a private ``torch.cuda.MemPool`` that lives as long as the program, so no
other allocation can land in a buffer the recorded launches point at.  No CPU fallback: recording needs a ROCm GPU.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from .. import _abi
from . import graphs as _graphs
from .graphs import _flatten, _rebuild


class NotReplayable(RuntimeError):
    pass


import os as _os  # noqa: E402

#: DT_REC_CHECK=0 skips the two replay checks of a recording (experiment switch of the round-6 fault hunt; default on)
_USE_CHECK = _os.environ.get("DT_REC_CHECK", "1") != "0"


class _LaunchGuard(TorchDispatchMode):
    """Collects the torch ops that produced or modified GPU data during a recording pass (anything that is neither an
    allocation nor a view)."""

    _ALLOC = ("empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided")

    def __init__(self):
        super().__init__()
        self.offenders = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name in self._ALLOC:
            return out
        flat_in, flat_out = [], []
        _flatten((args, kwargs or {}), flat_in)
        _flatten(out, flat_out)
        gpu_out = [t for t in flat_out if t.is_cuda]
        gpu_in = [t for t in flat_in if t.is_cuda]
        if not gpu_out and not gpu_in:
            return out
        mutable = bool(getattr(getattr(func, "_schema", None), "is_mutable", False))
        stores = {t.untyped_storage().data_ptr() for t in gpu_in}
        fresh = [t for t in gpu_out if t.numel() and t.untyped_storage().data_ptr() not in stores]
        if mutable or fresh:
            self.offenders.append(str(func))
        return out


class RecordedCallable:
    def __init__(self, fn, warmup=2, between=None, cut_config=None, check=True):
        """between(tag): called during replay after every segment that a ``graphs.cut(tag)`` call in fn ended.
        cut_config(): hashable describing which cuts fn WOULD make right now and which process-wide settings its kernel
        choices depend on; part of the cache key.  check: replay once right after recording and compare with the recording
        pass's own (eagerly executed) results bit for bit."""
        self.fn = fn
        self.warmup = max(1, int(warmup))
        self.between = between
        self.cut_config = cut_config
        self.check = check and _USE_CHECK
        self._entries = {}
        self.recordings = 0
        self.replays = 0
        self._tags = None

    # graphs.cut(tag) lands here while this object records (graphs._TLS.capturing is self)
    def _cut(self, tag):
        if _abi.lib().dt_program_mark() < 0:
            _abi.check(1, "dt_program_mark")
        self._tags.append(tag)

    def _record(self, args, kwargs, tensors, stream):
        L = _abi.lib()
        dev = tensors[0].device
        # own copies of the inputs for the recording: never aliased, whatever the caller handed over
        static = [t.detach().clone(memory_format=torch.preserve_format) for t in tensors]
        s_args, s_kwargs = _rebuild((args, kwargs), iter(static))
        _graphs._TLS.warming = True  # (one-shot hooks of fn are not consumed by throw-away passes)
        try:
            for _ in range(self.warmup):
                self.fn(*s_args, **s_kwargs)
                torch.cuda.current_stream(dev).synchronize()  # weight packs / library scratch exist and are complete
        finally:
            _graphs._TLS.warming = False
        pool = torch.cuda.MemPool()
        self._tags = []
        guard = _LaunchGuard()
        _abi.check(L.dt_program_begin(stream), "dt_program_begin")
        prog = C.c_void_p()
        try:
            for t in static:
                if t.numel() and L.dt_program_input(t.data_ptr(), t.untyped_storage().nbytes() - (t.data_ptr() - t.untyped_storage().data_ptr())) < 0:
                    _abi.check(1, "dt_program_input")
            _graphs._TLS.capturing = self
            _graphs._TLS.recording = True
            try:
                with torch.cuda.use_mem_pool(pool, device=dev), guard:
                    out = self.fn(*s_args, **s_kwargs)
            finally:
                _graphs._TLS.capturing = None
                _graphs._TLS.recording = False
        except BaseException:
            L.dt_program_abort()
            raise
        _abi.check(L.dt_program_end(C.byref(prog)), "dt_program_end")
        tags, self._tags = list(self._tags), None
        if guard.offenders:
            L.dt_program_free(prog)
            raise NotReplayable("torch ops launched kernels of their own inside a recorded step (they would be missing from every "
                                "replay): " + ", ".join(sorted(set(guard.offenders))))
        ent = dict(prog=prog, pool=pool, tags=tags, out=out, nseg=int(L.dt_program_info(prog, 1)), static=static,
                   slots=[i for i, t in enumerate(static) if t.numel()], stream=stream.value,
                   launches=int(L.dt_program_info(prog, 0)), patches=int(L.dt_program_info(prog, 2)),
                   lookalikes=int(L.dt_program_info(prog, 5)))
        if ent["lookalikes"] and _os.environ.get("DT_BENCH_TRACE"):
            import sys as _sys

            print(f"[program] {ent['lookalikes']} non-pointer argument word(s) hold values inside an input range (not patched)",
                  file=_sys.stderr, flush=True)
        ent["ptrs"] = (C.c_void_p * max(1, len(ent["slots"])))()
        self.recordings += 1
        if self.check:
            outs = []
            _flatten(out, outs)
            want = [o.clone() for o in outs]
            for o in outs:
                if o.is_floating_point():
                    o.fill_(float("nan"))
                else:
                    o.zero_()
            same = lambda a, b: torch.equal(a, b) or (a.is_floating_point() and torch.equal(torch.nan_to_num(a, nan=12345.0),
                                                                                            torch.nan_to_num(b, nan=12345.0)))
            self._launch(ent, static, stream, run_between=False)
            torch.cuda.current_stream(dev).synchronize()
            bad = [i for i, (a, b) in enumerate(zip(want, outs)) if not same(a, b)]
            if bad:
                L.dt_program_free(prog)
                raise NotReplayable(f"replay of the recorded step differs from its eager execution in outputs {bad} "
                                    f"({ent['launches']} launches recorded): something in the step did not go through the library")
            # relocation check: the same input VALUES at other addresses, with the recorded addresses poisoned -- a recorded
            # pointer into an input that the patch table missed would read the poison, a non-pointer word that the table
            # caught by accident would now change
            moved = [t.clone(memory_format=torch.preserve_format) for t in static]
            keep = [t.clone(memory_format=torch.preserve_format) for t in static]
            for t in static:
                if t.is_floating_point():
                    t.fill_(float("nan"))
                elif t.dtype != torch.bool:
                    t.fill_(-1)
            for o in outs:
                if o.is_floating_point():
                    o.fill_(float("nan"))
                else:
                    o.zero_()
            self._launch(ent, moved, stream, run_between=False)
            torch.cuda.current_stream(dev).synchronize()
            for t, k in zip(static, keep):
                t.copy_(k)
            bad = [i for i, (a, b) in enumerate(zip(want, outs)) if not same(a, b)]
            if bad:
                L.dt_program_free(prog)
                raise NotReplayable(f"replay on relocated inputs differs in outputs {bad}: the patch table of the program "
                                    f"({ent['patches']} patches for {len(ent['slots'])} inputs) does not cover every use of an input")
        return ent

    def _launch(self, ent, tensors, stream, run_between=True):
        L = _abi.lib()
        ptrs = ent["ptrs"]
        for j, i in enumerate(ent["slots"]):
            ptrs[j] = tensors[i].data_ptr()
        n = len(ent["slots"])
        prog, tags, nseg = ent["prog"], ent["tags"], ent["nseg"]
        if nseg == 1:
            rc = L.dt_program_launch(prog, -1, ptrs, n, stream)
            if rc:
                _abi.check(rc, "dt_program_launch")
            return
        for seg in range(nseg):
            rc = L.dt_program_launch(prog, seg, ptrs, n, stream)
            if rc:
                _abi.check(rc, "dt_program_launch")
            if run_between and seg < len(tags) and self.between is not None:
                self.between(tags[seg])

    def __call__(self, *args, **kwargs):
        tensors = []
        sig = _flatten((args, kwargs), tensors)
        if not tensors or not tensors[0].is_cuda:
            raise RuntimeError("RecordedCallable needs ROCm GPU tensors (no CPU fallback)")
        stream = _abi.current_stream(tensors[0].device)
        key = (sig, stream.value, self.cut_config() if self.cut_config is not None else None)
        ent = self._entries.get(key)
        if ent is None:
            ent = self._entries[key] = self._record(args, kwargs, tensors, stream)
        self._launch(ent, tensors, stream)
        self.replays += 1
        return ent["out"]

    def info(self):
        """One dict per recorded program: launches, segments, patches, input slots."""
        return [dict(launches=e["launches"], segments=e["nseg"], patches=e["patches"], inputs=len(e["slots"]), stream=e["stream"],
                     lookalikes=e["lookalikes"])
                for e in self._entries.values()]

    def reset(self):
        L = _abi.lib()
        for e in self._entries.values():
            L.dt_program_free(e["prog"])
        self._entries.clear()

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass
