"""bench.py --gpus N without a launcher must start its own N ranks (the driver invokes it as plain `python bench.py --gpus N`).
CPU only: the subprocess call is intercepted."""
import importlib
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_builds_a_torchrun_command(monkeypatch):
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    import subprocess

    import torch

    calls = {}

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env

        class R:
            returncode = 0

        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.self_launch(4) == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(REPO, "bench.py")):]
    assert tail[1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # asking for more GPUs than the node has fails loudly instead of hanging in rendezvous
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.self_launch(2) == 2


def test_main_rejects_inconsistent_world(monkeypatch, capsys):
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value)
