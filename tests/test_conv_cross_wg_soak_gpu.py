"""GPU: short form of scripts/stress_cross_wg.py inside the suite: the cross-workgroup K reduction of
csrc/conv.hip orders its partial-block payload against the arrival counter with write-through stores + vmcnt(0) + a relaxed
device-scope atomic -- outside the HIP memory model, so a compiler, firmware or driver change could break it silently.
6000 launches of the six layer shapes that use it, from two streams with L2-evicting traffic in between, every output
compared bit for bit (the 300 000-launch run is profiles/r2s_stress_cross_wg.txt)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_cross_workgroup_reduction_soak_short():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "scripts"))
    import stress_cross_wg

    assert stress_cross_wg.run(6000, verbose=False) == 0
