"""GPU_MAX_HW_QUEUES for processes that keep several keyframes in flight (parallel.KeyframePipeline).

The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With 4 lane streams plus
the default stream a lane shares its queue with another stream and the overlap the lanes exist for is partly lost (647 instead
of 742 frames/s at 640x480, profiles/r5_hwq_probe.txt).  The variable is read when the runtime initialises, i.e. at the
first HIP call of the process: ``ensure()`` must run BEFORE torch is imported.  This module imports nothing heavy.
"""
from __future__ import annotations

import os
import sys
import warnings

VAR = "GPU_MAX_HW_QUEUES"
DEFAULT_QUEUES = 4  # the runtime's default when the variable is unset
WANT = 8


def ensure(in_flight, want=WANT):
    """Set GPU_MAX_HW_QUEUES (unless the environment already does) when ``in_flight`` lanes plus the default stream exceed the
    runtime's default of four queues.  Returns the value in force, as a string, or None when nothing is set."""
    if int(in_flight) + 1 > DEFAULT_QUEUES:
        if VAR not in os.environ and "torch" in sys.modules:
            try:
                import torch

                late = torch.cuda.is_initialized()
            except Exception:
                late = False
            if late:
                warnings.warn(f"{VAR} requested after the HIP runtime initialised: it has no effect in this process", stacklevel=2)
        os.environ.setdefault(VAR, str(int(want)))
    return os.environ.get(VAR)


def check(in_flight):
    """Warn when ``in_flight`` lanes + the default stream do not fit the hardware queues of this process."""
    have = int(os.environ.get(VAR, DEFAULT_QUEUES) or DEFAULT_QUEUES)
    if int(in_flight) + 1 > have:
        warnings.warn(f"{in_flight} keyframes in flight + the default stream on {have} hardware queues ({VAR}): lanes will share "
                      f"queues; call doubletake_amd.hwqueues.ensure({in_flight}) before importing torch", stacklevel=3)
        return False
    return True
