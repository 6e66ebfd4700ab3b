"""world_size-2 CPU (gloo) test of the keyframe-shard exchange: canonical order, payload integrity,
replica determinism.  The GPU fuser is replaced by a recording callback (the collective logic is
device independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from doubletake_amd import parallel as par


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, w = 6, 8
    log = []
    fuser = par.KeyframeShardFuser(torch.device("cpu"), world, rank, 2 * h, 2 * w,
                                   fuse_fn=lambda d, K, T: log.append((d.clone(), K.clone(), T.clone())), pool=16)
    for step in range(3):
        depth = torch.full((1, 1, h, w), float(10 * step + rank + 1)) + torch.arange(w).float() * 0.01
        n = fuser.exchange_and_fuse(depth, step)
        assert n == world
    torch.save(log, os.path.join(out_dir, f"log{rank}.pt"))
    dist.destroy_process_group()


def test_exchange_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    logs = [torch.load(os.path.join(tmp_path, f"log{r}.pt")) for r in range(world)]
    assert len(logs[0]) == len(logs[1]) == 3
    for step in range(3):
        d0, K0, T0 = logs[0][step]
        d1, K1, T1 = logs[1][step]
        # every replica sees the same frames in the same (rank-major) order
        assert torch.equal(d0, d1) and torch.equal(K0, K1) and torch.equal(T0, T1)
        assert d0.dtype == torch.float16 and tuple(d0.shape) == (2, 1, 6, 8)
        for r in range(world):
            want = (torch.full((6, 8), float(10 * step + r + 1)) + torch.arange(8).float() * 0.01).half()
            assert torch.equal(d0[r, 0], want)
    # the pool index follows the global frame id: step*world + rank
    from doubletake_amd.utils import synthetic as syn

    _, K, T = syn.tsdf_frames(16, 6, 8, seed=5, bounds=par.KeyframeShardFuser.BOUNDS)
    for step in range(3):
        for r in range(world):
            np.testing.assert_array_equal(logs[0][step][1][r].numpy(), K[(step * world + r) % 16].astype(np.float16))
            np.testing.assert_array_equal(logs[0][step][2][r].numpy(), T[(step * world + r) % 16].astype(np.float16))


def test_pack_roundtrip_and_sharding():
    d = torch.rand(3, 1, 5, 7)
    K = torch.rand(3, 4, 4)
    T = torch.rand(3, 4, 4)
    dd, KK, TT = par.unpack_update(par.pack_update(d, K, T), 5, 7)
    assert torch.equal(dd, d.half()) and torch.equal(KK, K.half()) and torch.equal(TT, T.half())
    shards = [par.shard_keyframes(11, 4, r) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(11))
    assert shards[1] == [1, 5, 9]
