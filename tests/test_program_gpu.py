"""GPU: launch programs (include/doubletake_hip.h dt_program_*, csrc/program.hip, utils/program.py): the kernel launches of a
step recorded at the C ABI and replayed by one call -- at the ABI itself, through RecordedCallable, and through the model's
``enable_launch_programs`` (bit-identical to the eager entry points, input addresses patched, weights and settings keyed)."""
import ctypes as C

import numpy as np
import pytest
import torch

from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _model(h=24, w=32, k=3, D=16, dec="skip", b=1, seed=5):
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint

    enc, widths = ("resnet18d", [64, 64, 128, 256, 512]) if dec == "skip" else ("efficientnet", [24, 48, 64, 160, 256])
    model = DepthModelCVHint(4 * h, 4 * w, image_encoder_name=enc, depth_decoder_name=dec, matching_num_depth_bins=D,
                             model_num_views=k + 1, matching_encoder_type=None)
    gu.set_formula_weights(model, seed)
    model = model.to(gu.dev())

    def frame(s):
        t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, s))
        pyr = [torch.from_numpy(p).to(gu.dev()).contiguous(memory_format=torch.channels_last)
               for p in syn.prior_pyramid(b, widths, 2 * h, 2 * w, s + 50)]
        return t, pyr

    def run(fr):
        t, pyr = fr
        return model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                           t["cur_invK"], gu.hint_dict(t), return_mask=True)

    return model, frame, run


def test_program_abi_records_patches_and_replays():
    """The C ABI alone: record two library launches (exp of x into y, exp of y into z), replay them on another input address,
    in segments, and check the refusals (other stream, wrong slot count, overlapping inputs, stale handle)."""
    import gpu_util as gu
    from doubletake_amd import _abi

    L = _abi.lib()
    dev = gu.dev()
    st = _abi.current_stream(dev)
    x1 = torch.linspace(-1, 1, 1000, device=dev)
    x2 = torch.linspace(0, 2, 1000, device=dev)
    y, z = torch.empty_like(x1), torch.empty_like(x1)
    _abi.check(L.dt_program_begin(st), "begin")
    assert L.dt_program_input(x1.data_ptr(), x1.numel() * 4) == 0
    assert L.dt_program_input(x1.data_ptr() + 16, 64) == -1 and b"overlap" in L.dt_last_error()
    c0 = L.dt_kernel_launch_count()
    _abi.check(L.dt_exp_f32(x1.data_ptr(), y.data_ptr(), 1000, st), "exp")
    assert L.dt_program_mark() == 1
    _abi.check(L.dt_exp_f32(y.data_ptr(), z.data_ptr(), 1000, st), "exp")
    prog = C.c_void_p()
    _abi.check(L.dt_program_end(C.byref(prog)), "end")
    assert L.dt_kernel_launch_count() - c0 == 2          # recording launches execute
    torch.cuda.synchronize()
    assert torch.equal(z, torch.exp(torch.exp(x1))) or float((z - torch.exp(torch.exp(x1))).abs().max()) < 1e-5
    assert [L.dt_program_info(prog, i) for i in range(4)] == [2, 2, 1, 1]
    ptrs = (C.c_void_p * 1)(x2.data_ptr())
    y.zero_(), z.zero_()
    _abi.check(L.dt_program_launch(prog, 0, ptrs, 1, st), "launch seg 0")   # first segment only, on the OTHER input
    torch.cuda.synchronize()
    want_y = torch.empty_like(x2)
    _abi.check(L.dt_exp_f32(x2.data_ptr(), want_y.data_ptr(), 1000, st), "exp")
    torch.cuda.synchronize()
    assert torch.equal(y, want_y) and float(z.abs().max()) == 0.0
    _abi.check(L.dt_program_launch(prog, 1, ptrs, 1, st), "launch seg 1")
    torch.cuda.synchronize()
    want_z = torch.empty_like(x2)
    _abi.check(L.dt_exp_f32(want_y.data_ptr(), want_z.data_ptr(), 1000, st), "exp")
    torch.cuda.synchronize()
    assert torch.equal(z, want_z)
    ptrs[0] = x1.data_ptr()
    _abi.check(L.dt_program_launch(prog, -1, ptrs, 1, st), "launch all")
    torch.cuda.synchronize()
    assert float((z - torch.exp(torch.exp(x1))).abs().max()) < 1e-5
    # refusals
    other = torch.cuda.Stream(dev)
    assert L.dt_program_launch(prog, -1, ptrs, 1, C.c_void_p(other.cuda_stream)) != 0 and b"recorded on" in L.dt_last_error()
    assert L.dt_program_launch(prog, -1, ptrs, 0, st) != 0
    assert L.dt_program_launch(prog, 5, ptrs, 1, st) != 0
    assert L.dt_program_free(prog) == 0
    assert L.dt_program_launch(prog, -1, ptrs, 1, st) != 0 and b"live" in L.dt_last_error()
    assert L.dt_program_free(prog) != 0
    assert L.dt_program_end(C.byref(prog)) != 0            # nothing is being recorded
    # launches on another stream are not recorded
    _abi.check(L.dt_program_begin(st), "begin")
    with torch.cuda.stream(other):
        _abi.check(L.dt_exp_f32(x1.data_ptr(), y.data_ptr(), 1000, _abi.current_stream(dev)), "exp")
    _abi.check(L.dt_program_end(C.byref(prog)), "end")
    assert L.dt_program_info(prog, 0) == 0
    L.dt_program_free(prog)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dec,b", [("skip", 1), ("skip", 2), ("unet_pp", 1)])
def test_model_launch_program_is_bit_identical_to_eager(dec, b):
    """enable_launch_programs: every output of forward_from_features equals the eager path's bit for bit, on the recording frame
    and on other frames (their addresses are patched in), for the small and the full decoder and batch 2."""
    model, frame, run = _model(dec=dec, b=b)
    frames = [frame(30 + i) for i in range(3)]
    want = [{k: v.clone() for k, v in run(f).items()} for f in frames]
    model.enable_launch_programs(True)
    for rep in range(2):
        for f, w in zip(frames, want):
            got = run(f)
            torch.cuda.synchronize()
            assert set(got) == set(w)
            for k in w:
                assert torch.equal(got[k], w[k]), (dec, b, rep, k)
    info = model._recorded_forward.info()
    assert len(info) == 1 and info[0]["segments"] == 4 and info[0]["launches"] >= 20 and info[0]["patches"] >= info[0]["inputs"] >= 10
    assert model._recorded_forward.recordings == 1 and model._recorded_forward.replays == 6
    model.enable_launch_programs(False)


def test_launch_program_follows_weights_settings_and_hooks():
    """A recorded program is keyed on the weights (version and address), on the process-wide launch settings and on the stream;
    the event hook and the one-shot after_volume hook fire at their cut points on replay."""
    from doubletake_amd.modules import conv_ops as ops
    from doubletake_amd.modules.cost_volume import FeatureVolumeManager

    model, frame, run = _model()
    f = frame(40)
    base = {k: v.clone() for k, v in run(f).items()}
    model.enable_launch_programs(True)
    rc = model._recorded_forward
    assert torch.equal(run(f)["depth_pred_s0_b1hw"], base["depth_pred_s0_b1hw"]) and rc.recordings == 1
    # weights change in place -> new recording, new result = the eager result with the new weights
    with torch.no_grad():
        model.depth_decoder.out4[4].bias.add_(0.25)
    got = run(f)["log_depth_pred_s0_b1hw"].clone()
    assert rc.recordings == 2 and model._recorded_forward is rc   # (reset() keeps the object, drops the stale programs)
    assert float((got - (base["log_depth_pred_s0_b1hw"] + 0.25)).abs().max()) < 1e-6
    # plan objective flipped -> another program; flipped back -> the first one is valid again (no third recording)
    n0 = rc.recordings
    prev = ops.current_plan_objective()
    try:
        ops.set_plan_objective(ops.PLAN_THROUGHPUT)
        a = run(f)["depth_pred_s0_b1hw"].clone()
        assert rc.recordings == n0 + 1
    finally:
        ops.set_plan_objective(prev)
    b_ = run(f)["depth_pred_s0_b1hw"].clone()
    assert rc.recordings == n0 + 1 and float((a - b_).abs().max()) < 1e-4
    # another stream -> its own program and output buffers
    side = torch.cuda.Stream(f[0]["cur_feats"].device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = run(f)["depth_pred_s0_b1hw"]
    side.synchronize()
    assert rc.recordings == n0 + 2 and torch.equal(c, b_) and c.data_ptr() != b_.data_ptr()
    # hooks at the cut points
    tags, shots = [], []
    FeatureVolumeManager._event_hook = staticmethod(tags.append)
    try:
        model.after_volume = lambda: shots.append(1)
        run(f)
        run(f)
    finally:
        FeatureVolumeManager._event_hook = None
    assert tags == ["mlp_begin", "mlp_end"] * 2 and shots == [1] and rc.recordings == n0 + 2
    model.enable_launch_programs(False)


def test_launch_program_refuses_steps_with_foreign_kernels():
    """A torch op that launches its own kernel inside the recorded step would be missing from every replay: recording raises
    NotReplayable and names it (here: fp64 matching features, which the eager path converts with a torch cast)."""
    from doubletake_amd.utils.program import NotReplayable, RecordedCallable

    model, frame, run = _model()
    t, pyr = frame(41)
    t = dict(t, cur_feats=t["cur_feats"].double())
    model.enable_launch_programs(True)
    with pytest.raises(NotReplayable, match="_to_copy|copy"):
        run((t, pyr))
    model.enable_launch_programs(False)
    # ... and RecordedCallable on its own: an output that silently depends on a foreign kernel
    rc = RecordedCallable(lambda x: x * 2.0)
    with pytest.raises(NotReplayable):
        rc(torch.ones(8, device=pyr[0].device))


@pytest.mark.parametrize("volume_type", ["mlp_feature_volume", "simple_cost_volume"])
def test_simplerecon_model_types_replay_from_launch_programs_too(volume_type):
    """DepthModel (SimpleRecon, reference experiment_modules/sr_depth_model.py:186-204) shares the hot-path base class: its
    metadata-MLP volume (no hints) and its dot-product volume replay from launch programs bit-identically as well."""
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModel

    h, w, k, D = 24, 32, 3, 16
    model = DepthModel(4 * h, 4 * w, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=k + 1,
                       matching_encoder_type=None, feature_volume_type=volume_type)
    gu.set_formula_weights(model, 11)
    model = model.to(gu.dev())

    def frame(s):
        t = gu.to_dev(syn.volume_inputs(1, k, h, w, 16, s))
        pyr = [torch.from_numpy(p).to(gu.dev()).contiguous(memory_format=torch.channels_last)
               for p in syn.prior_pyramid(1, [64, 64, 128, 256, 512], 2 * h, 2 * w, s + 50)]
        return t, pyr

    def run(fr):
        t, pyr = fr
        return model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                           t["cur_invK"], None, return_mask=True)

    frames = [frame(70 + i) for i in range(2)]
    want = [{kk: v.clone() for kk, v in run(f).items() if v is not None} for f in frames]
    model.enable_launch_programs(True)
    for f, wnt in zip(frames + frames, want + want):
        got = run(f)
        torch.cuda.synchronize()
        for kk in wnt:
            assert torch.equal(got[kk], wnt[kk]), (volume_type, kk)
    assert model._recorded_forward.recordings == 1 and model._recorded_forward.info()[0]["lookalikes"] == 0
    model.enable_launch_programs(False)
