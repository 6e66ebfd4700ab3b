"""bench.py prints exactly one JSON line with the driver's contract fields (short run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0.3 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None or rf["traffic"] > 1e6
    dot = d["roofline_warp_match_dot"]
    assert dot["bound"] == "hbm" and dot["unit"] == "GB/s" and dot["peak"] == 8000.0
