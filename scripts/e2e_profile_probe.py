#!/usr/bin/env python3
"""Run only bench.end_to_end (model("test", ...) with the matching encoder) -- for a kernel trace of that path:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e -o e2e -- python scripts/e2e_profile_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

import bench

dev = torch.device("cuda:0")
inp, pyr, t, pyr_t = bench.build_inputs(dev, 1000)
model = bench.build_model(dev)
print(json.dumps(bench.end_to_end(dev, t, pyr_t, model, frames=int(os.environ.get("DT_FRAMES", "20")))))
