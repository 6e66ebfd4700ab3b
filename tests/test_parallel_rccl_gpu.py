"""GPU: the keyframe-shard exchange over a real RCCL process group (world size 1 -- the box has one GPU) with the real HIP
fuser: init, all_gather_into_tensor of the packed fp16 update on the device, unpack, replica integrate.  The replica must
be bit-identical to fusing the same frames serially without any collective."""
import os
import socket

import numpy as np
import pytest
import torch

from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

BD = dict(xmin=-1.28, xmax=1.28, ymin=-1.12, ymax=1.12, zmin=0.0, zmax=2.24)


def test_world1_rccl_exchange_equals_serial_fuse():
    import torch.distributed as dist

    import gpu_util as gu
    from doubletake_amd import parallel as par
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    torch.cuda.set_device(dev)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    os.environ.pop("NCCL_DEBUG", None)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        H, W = 120, 160
        depth, K, T = syn.tsdf_frames(6, H, W, seed=3, bounds=BD)
        depth = depth * np.float32(0.6)
        d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
        serial = OurFuser(None, 0.04, 3.0, bounds=BD)
        replica = OurFuser(None, 0.04, 3.0, bounds=BD)
        sf = par.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=replica, force_collective=True)
        # batches of 2, 2, 1, 1 frames: the ragged path (rows > count) goes through the collective as well
        sizes = [2, 2, 1, 1]
        starts = np.cumsum([0] + sizes)
        n = par.run_sharded_pass(len(sizes), lambda i: sizes[i],
                                 lambda i: (d[starts[i]:starts[i + 1]], k[starts[i]:starts[i + 1]], t[starts[i]:starts[i + 1]]), sf)
        for i in range(len(sizes)):
            sl = slice(starts[i], starts[i + 1])
            serial.fuse_frames(d[sl], k[sl], t[sl], None)
        torch.cuda.synchronize()
        assert n == 6 and sf.frames_fused == 6 and sf._all is not None and sf._all.is_cuda
        a, b = serial.tsdf_fuser_pred.tsdf, replica.tsdf_fuser_pred.tsdf
        assert (a.tsdf_weights > 0).sum().item() > 10000
        assert torch.equal(a.tsdf_values.view(torch.int16), b.tsdf_values.view(torch.int16))
        assert torch.equal(a.tsdf_weights.view(torch.int16), b.tsdf_weights.view(torch.int16))
        assert torch.equal(a.voxel_bitmap, b.voxel_bitmap)
        # variable-size TSDF gather (scene-shard mode) over the same group
        got = {}
        par.run_scene_sharded([5], lambda s_: replica, 1, 0, on_scene_done=lambda s_, dd: got.__setitem__(s_, dd), device=dev)
        assert torch.equal(got[0]["tsdf_values"], b.tsdf_values) and torch.equal(got[0]["tsdf_weights"], b.tsdf_weights)
        np.testing.assert_array_equal(got[0]["origin_f32"], b.origin_f32)
    finally:
        dist.destroy_process_group()


def test_cfg5_size_two_pass_loop_through_the_collective_path():
    """BASELINE configs[4] loop at its own size on one GPU with the RCCL exchange forced (world 1): portrait 384x512 frames,
    96 planes, 7 source views, keyframe batches of 2 through loops.two_pass_fns + parallel.run_two_pass -- pass 1 with empty
    hints, marching cubes, pass 2 with rendered hints -- every TSDF update travelling through all_gather_into_tensor.
    The replicas must equal a run of the same loop without any collective, bit for bit."""
    import torch.distributed as dist

    import gpu_util as gu
    import test_drivers_gpu as drv
    from doubletake_amd import loops
    from doubletake_amd import parallel as par
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    torch.cuda.set_device(dev)
    H, W, k, D, b, nb = 512, 384, 7, 96, 2, 3
    H2, W2 = H // 2, W // 2
    bd = drv.BD
    model = drv._model(H, W, k, D, dev)
    cams = drv._cams(b * nb, H2, W2)
    surface, _, _ = syn.tsdf_frames(1, H2, W2, seed=3, bounds=bd)
    base = torch.from_numpy(surface[0:1] * np.float32(0.55)).to(dev)
    cover = []

    def model_fn(cur_data, src_data):
        out = model("test", cur_data, src_data, return_mask=True)
        assert tuple(out["depth_pred_s0_b1hw"].shape) == (b, 1, H2, W2)
        out["depth_pred_s0_b1hw"] = base + 0.02 * torch.tanh(out["depth_pred_s0_b1hw"] - 1.0)
        cover.append(float(cur_data["depth_hint_mask_b1hw"].mean()))
        return out

    load = lambda i: drv._batch(i, b, k, H, W, dev, cams)

    def run(force):
        first, between, second = loops.two_pass_fns(model_fn, load, (H2, W2), fuse_size=(H, W))
        hint_fuser, final_fuser = OurFuser(None, 0.04, 3.0, bounds=bd), OurFuser(None, 0.04, 3.0, bounds=bd)
        sf_h = par.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=hint_fuser, force_collective=force)
        sf_f = par.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=final_fuser, force_collective=force)
        n = par.run_two_pass(nb, lambda i: b, first, second, sf_h, sf_f, between_passes=between)
        torch.cuda.synchronize()
        return n, hint_fuser.tsdf_fuser_pred.tsdf, final_fuser.tsdf_fuser_pred.tsdf, sf_f

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    os.environ.pop("NCCL_DEBUG", None)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n_c, hint_c, final_c, sf = run(True)
        assert sf._all is not None and sf._all.is_cuda and tuple(sf._all.shape) == (b, H * W + 32)   # really went through the gather
    finally:
        dist.destroy_process_group()
    n_s, hint_s, final_s, _ = run(False)
    assert n_c == n_s == (b * nb, b * nb)
    assert cover[:nb] == [0.0] * nb and min(cover[nb:2 * nb]) > 0.2
    for a, c in ((hint_c, hint_s), (final_c, final_s)):
        assert (a.tsdf_weights > 0).sum().item() > 5000
        assert torch.equal(a.tsdf_values.view(torch.int16), c.tsdf_values.view(torch.int16))
        assert torch.equal(a.tsdf_weights.view(torch.int16), c.tsdf_weights.view(torch.int16))


def test_world1_rccl_slab_mode_at_the_final_volume_size():
    """KeyframeShardFuser(mode="slab") through a real RCCL group (world 1, collective forced) at the
    drivers' final-volume size (0.02 m over the 8 x 8 x 3.2 m room: 400 x 400 x 160, 205 MB per pass): exchange,
    slab-restricted integrate, byte all_gather of values / weights / active bits -- equal to the serial fuse."""
    import torch.distributed as dist

    import gpu_util as gu
    from doubletake_amd import parallel as par
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    torch.cuda.set_device(dev)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    os.environ.pop("NCCL_DEBUG", None)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        room = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
        H, W = 240, 320
        depth, K, T = syn.tsdf_frames(6, H, W, seed=5, bounds=room)
        d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
        serial = OurFuser(None, 0.02, 3.0, bounds=room)
        slab = OurFuser(None, 0.02, 3.0, bounds=room)
        assert tuple(slab.tsdf_fuser_pred.tsdf.tsdf_values.shape) == (400, 400, 160)
        sf = par.KeyframeShardFuser(dev, 1, 0, (H, W), fuser=slab, force_collective=True, mode="slab")
        assert sf.slab == (0, 400)
        sizes = [2, 2, 2]
        starts = np.cumsum([0] + sizes)
        n = par.run_sharded_pass(len(sizes), lambda i: sizes[i],
                                 lambda i: (d[starts[i]:starts[i + 1]], k[starts[i]:starts[i + 1]], t[starts[i]:starts[i + 1]]), sf)
        for i in range(len(sizes)):
            sl = slice(starts[i], starts[i + 1])
            serial.fuse_frames(d[sl], k[sl], t[sl], None)
        torch.cuda.synchronize()
        assert n == 6 and sf.slabs_current
        a, b = serial.tsdf_fuser_pred.tsdf, slab.tsdf_fuser_pred.tsdf
        assert (a.tsdf_weights > 0).sum().item() > 100000
        assert torch.equal(a.tsdf_values.view(torch.int16), b.tsdf_values.view(torch.int16))
        assert torch.equal(a.tsdf_weights.view(torch.int16), b.tsdf_weights.view(torch.int16))
        assert torch.equal(a.voxel_bitmap, b.voxel_bitmap)
    finally:
        dist.destroy_process_group()
