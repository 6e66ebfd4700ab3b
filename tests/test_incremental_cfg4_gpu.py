"""GPU: the incremental (online) loop at BASELINE configs[3] size -- 512x384 image, 7 source views, 64 planes, batch 1:
fuse -> marching cubes -> hint render -> TSDF weight sampling -> mesh-hint volume -> CVEncoder -> SkipDecoderRegression ->
nearest upsample -> fuse, six frames through doubletake_amd.loops.run_incremental_scan (reference
test_incremental.py:172-372).  Checked: the model output of frame 3 (whose hint comes from the TSDF fused from frames 0-2)
against the torch-CPU oracle fed the same hint maps (<= 1e-3 abs depth); the final TSDF bit for bit against the numpy TSDF
oracle integrating the same fused depths; the hint invariants of the reference on every frame."""
import numpy as np
import pytest
import torch

from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

BD = dict(xmin=-2.0, xmax=2.0, ymin=-2.0, ymax=2.0, zmin=0.0, zmax=2.4)
H, W, K_SRC, D = 384, 512, 7, 64
h, w, H2, W2 = H // 4, W // 4, H // 2, W // 2
NFRAMES = 6


def _cameras():
    """A slowly translating camera inside the room; intrinsics at depth-prediction (s0) and full depth resolution."""
    _, K, T = syn.tsdf_frames(1, H2, W2, seed=3, bounds=BD)
    Ks0, cam_T_world = [], []
    pose0 = np.linalg.inv(T[0].astype(np.float64))
    for f in range(NFRAMES):
        pose = pose0.copy()
        pose[:3, 3] += np.array([0.02 * f, 0.012 * f, 0.0])
        cam_T_world.append(np.linalg.inv(pose).astype(np.float32))
        Ks0.append(K[0])
    Ks0 = np.stack(Ks0)
    Kfull = Ks0.copy()
    Kfull[:, :2] *= 2.0
    return Ks0, Kfull, np.stack(cam_T_world)


def test_incremental_loop_cfg4_size():
    import gpu_util as gu
    from doubletake_amd import loops
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.tools.fusers_helper import OurFuser
    from oracle import torch_cpu_ref as tref
    from oracle import tsdf_ref

    dev = gu.dev()
    model = DepthModelCVHint(H, W, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=K_SRC + 1,
                             matching_encoder_type=None)
    gu.set_formula_weights(model, 7)
    model = model.to(dev)
    Ks0, Kfull, cTw = _cameras()
    surface, _, _ = syn.tsdf_frames(NFRAMES, H2, W2, seed=3, bounds=BD)
    frames, feats = [], []
    for f in range(NFRAMES):
        inp = syn.volume_inputs(1, K_SRC, h, w, 16, 40 + f)
        pyr = syn.prior_pyramid(1, [64, 64, 128, 256, 512], H2, W2, 60 + f)
        feats.append((inp, pyr))
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        cur = {"K_s0_b44": tt(Ks0[f:f + 1]), "invK_s0_b44": tt(np.linalg.inv(Ks0[f:f + 1])), "cam_T_world_b44": tt(cTw[f:f + 1]),
               "world_T_cam_b44": tt(np.linalg.inv(cTw[f:f + 1]).astype(np.float32)), "K_full_depth_b44": tt(Kfull[f:f + 1]),
               "_t": gu.to_dev(inp), "_pyr": [tt(p) for p in pyr], "_f": f}
        frames.append((cur, {}))
    seen = {}

    def model_fn(cur, src):
        t = cur["_t"]
        out = model.forward_from_features(cur["_pyr"], t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"],
                                          t["src_Ks"], t["cur_invK"], cur, return_mask=True)
        out["raw_depth_pred_s0_b1hw"] = out["depth_pred_s0_b1hw"]
        # random-weight networks do not predict metric depth: fuse a plausible surface that still depends on the prediction
        # (so frame t's hint really is a function of the model outputs of the frames before it)
        base = torch.from_numpy(surface[0:1] * np.float32(0.55)).to(dev)
        out["depth_pred_s0_b1hw"] = base + 0.02 * torch.tanh(out["raw_depth_pred_s0_b1hw"] - 1.0)
        return out

    fused = []

    def on_frame(i, cur, out):
        hm = cur["depth_hint_mask_b_b1hw"]
        assert torch.equal(torch.isnan(cur["depth_hint_b1hw"]), ~hm)          # NaN exactly where the mask is off
        assert (cur["sampled_weights_b1hw"][~hm] == 0).all()
        if i > 0 and hm.any():
            assert (cur["sampled_weights_b1hw"][hm] >= 0.025).all()           # test_incremental.py:246
        seen[i] = dict(cover=float(hm.float().mean()), raw=out["raw_depth_pred_s0_b1hw"].cpu().numpy(),
                       hint={n: cur[n].cpu().numpy().astype(np.float32) for n in ("depth_hint_b1hw", "sampled_weights_b1hw",
                                                                                 "depth_hint_mask_b1hw")})
        up = torch.nn.functional.interpolate(out["depth_pred_s0_b1hw"], size=(H, W), mode="nearest")
        fused.append(up[0, 0].cpu().numpy())

    fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    timer = loops.FrameTimer()
    n = loops.run_incremental_scan(model_fn, fuser, frames, (H2, W2), fuse_size=(H, W), on_frame=on_frame, timer=timer)
    times = timer.per_frame()  # test_incremental.py:274-288: model_time / hint_time per frame (ms)
    assert len(times["model_time"]) == len(times["hint_time"]) == NFRAMES
    assert times["hint_time"][0] == 0.0 and min(times["hint_time"][1:]) > 0.0 and min(times["model_time"]) > 0.0
    torch.cuda.synchronize()
    assert n == NFRAMES
    # one observation stays below the 0.025 weight cut (tools/tsdf.py:546-549); from the third frame on the hint covers the view
    assert seen[0]["cover"] == 0.0 and seen[1]["cover"] == 0.0
    assert seen[3]["cover"] > 0.3 and seen[5]["cover"] > 0.3

    # ---- frame 3 against the oracle, with the hint maps the GPU loop produced -------------------------------------------
    inp, pyr = feats[3]
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    lin = lambda pre: [(sd[f"{pre}.net.{i}.weight"], sd[f"{pre}.net.{i}.bias"]) for i in (0, 2, 4)]
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    vol, planes = tref.hint_volume_loop(*(T_(inp[k]) for k in ("cur_feats", "src_feats", "src_extrinsics", "src_poses", "src_Ks",
                                                               "cur_invK", "min_depth", "max_depth")), D, lin("cost_volume.mlp"),
                                        hint={k: T_(v) for k, v in seen[3]["hint"].items()}, hint_mlp=lin("cost_volume.hint_mlp"))
    ref = tref.skip_decoder_regression([T_(pyr[0])] + tref.cv_encoder(vol, [T_(p) for p in pyr[1:]], sub("cost_volume_net.")),
                                       sub("depth_decoder."))
    want = np.exp(ref["log_depth_pred_s0_b1hw"].numpy())
    assert np.abs(seen[3]["raw"] - want).max() < 1e-3

    # ---- final TSDF bit for bit against the numpy oracle integrating the same depths ---------------------------------------
    vol_ref = tsdf_ref.TSDFVolume(BD, 0.04)
    for f in range(NFRAMES):
        tsdf_ref.integrate(vol_ref, fused[f], Kfull[f], cTw[f], max_depth=3.0)
    t = fuser.tsdf_fuser_pred.tsdf
    np.testing.assert_array_equal(t.tsdf_values.cpu().numpy().view(np.uint16), vol_ref.values.view(np.uint16))
    np.testing.assert_array_equal(t.tsdf_weights.cpu().numpy().view(np.uint16), vol_ref.weights.view(np.uint16))
    keys = t.active_keys().cpu().numpy()
    assert {tuple(k) for k in keys.tolist()} == {tuple(int(v) for v in k) for k in vol_ref.active}


def test_get_fuser_factory_and_view_helpers():
    """tools/fusers_helper.py:214-243 and utils/generic_utils.py:111-137."""
    from types import SimpleNamespace

    import gpu_util as gu
    from doubletake_amd.tools import fusers_helper
    from doubletake_amd.utils import generic_utils as G

    opts = SimpleNamespace(dataset="scannet", dataset_path="/nonexistent", split="test", depth_fuser="ours", fusion_resolution=0.08,
                           fusion_max_depth=3.0, fuse_color=False, extended_neg_truncation=True)
    f = fusers_helper.get_fuser(opts, "scene0707_00")
    assert isinstance(f, fusers_helper.OurFuser) and f.extended_neg_truncation
    assert tuple(f.tsdf_fuser_pred.tsdf.tsdf_values.shape) == (256, 256, 256)   # +-10 m default at 0.08 m
    with pytest.raises(ValueError):
        fusers_helper.get_fuser(SimpleNamespace(**{**vars(opts), "depth_fuser": "nope"}), "s")
    x = torch.arange(2 * 3 * 4 * 5, device=gu.dev()).view(6, 4, 5)
    assert G.tensor_B_to_bM(x, 2, 3).data_ptr() == x.data_ptr()   # views, also on the device
    m = f.get_mesh()
    assert m.vertices.shape[1] == 3


def test_hint_sampling_fp32_vs_half_model_differ_only_at_the_weight_cut():
    """the product samples the fused weight volume with an fp32 blend of the half voxels (the reference's
    CPU branch, pinned by the goldens); on a GPU the reference's grid_sample runs in half (tools/tsdf.py:327-330).  With
    the opt-in half model (SAMPLE_FP16_MATH / fp16_math=True: grid, result and cut rounded to half) the hint mask of the
    incremental loop at cfg4 size may only change for pixels whose sampled weight lies within half precision of the 0.025
    cut, and only for a small share of the image; the kept weights agree to half precision."""
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.utils.rendering_utils import prepare_mesh_hint_fused

    dev = gu.dev()
    Ks0, Kfull, cTw = _cameras()
    surface, _, _ = syn.tsdf_frames(NFRAMES, H2, W2, seed=3, bounds=BD)
    fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    stats = []
    for f in range(NFRAMES):
        cur = {"K_s0_b44": tt(Ks0[f:f + 1]), "invK_s0_b44": tt(np.linalg.inv(Ks0[f:f + 1])), "cam_T_world_b44": tt(cTw[f:f + 1]),
               "world_T_cam_b44": tt(np.linalg.inv(cTw[f:f + 1]).astype(np.float32))}
        if f >= 2:
            a, b = dict(cur), dict(cur)
            prepare_mesh_hint_fused(fuser, a, H2, W2, fp16_math=False)
            prepare_mesh_hint_fused(fuser, b, H2, W2, fp16_math=True)
            ma, mb = a["depth_hint_mask_b_b1hw"], b["depth_hint_mask_b_b1hw"]
            # sampled weights BEFORE the cut, fp32 branch (what decides the mask)
            c = dict(cur)
            prepare_mesh_hint_fused(fuser, c, H2, W2, weight_threshold=float("-inf"), fp16_math=False)
            sw = c["sampled_weights_b1hw"]
            flip = ma ^ mb
            both = ma & mb
            n_flip, n_kept = int(flip.sum()), int(ma.sum())
            stats.append((f, n_kept, n_flip, float((sw[flip] - 0.025).abs().max()) if n_flip else 0.0,
                          float((a["sampled_weights_b1hw"][both] - b["sampled_weights_b1hw"][both]).abs().max())))
            assert n_kept > 0.3 * H2 * W2
            # flips only within half precision of the sampling position (0.05-0.1 voxel) of the cut ...
            assert n_flip == 0 or float((sw[flip] - 0.025).abs().max()) < 4e-3
            # ... and rare
            assert n_flip <= 0.01 * H2 * W2
            # where both keep the pixel the weights agree to what half positions / a half result allow
            assert float((a["sampled_weights_b1hw"][both] - b["sampled_weights_b1hw"][both]).abs().max()) < 8e-3
            # the hint depth itself does not depend on the sampling arithmetic
            assert torch.equal(a["depth_hint_b1hw"][both], b["depth_hint_b1hw"][both])
        d = tt(surface[0:1] * np.float32(0.55) + np.float32(0.004 * f))
        up = torch.nn.functional.interpolate(d, size=(H, W), mode="nearest")
        fuser.fuse_frames(up, tt(Kfull[f:f + 1]), tt(cTw[f:f + 1]), None)
    print("fp32 vs half-model hint sampling (frame, kept, flipped, max |w-0.025| of flips, max |dw| kept):", stats)


@pytest.mark.parametrize("programs", [False, True])
def test_incremental_scans_in_flight_equal_the_same_scans_alone(programs):
    """loops.run_incremental_scans (round 6): three scans of the incremental mode in flight on HIP-stream lanes, one shared
    model (with and without launch programs).  Scans are independent, the frames of a scan are not: every scan's predicted
    depths and its final TSDF (values, weights, active bits) must equal, bit for bit, what the scan gives when it runs alone."""
    import gpu_util as gu
    from doubletake_amd import loops
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    hh, ww, k, D2 = 48, 64, 3, 16
    Hh, Wh = 2 * hh, 2 * ww
    model = DepthModelCVHint(4 * hh, 4 * ww, depth_decoder_name="skip", matching_num_depth_bins=D2, model_num_views=k + 1,
                             matching_encoder_type=None)
    gu.set_formula_weights(model, 9)
    model = model.to(dev)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lengths = (5, 3, 4)

    def make_scan(s, record):
        surface, K, T = syn.tsdf_frames(lengths[s], Hh, Wh, seed=11 + s, bounds=BD)
        batches = []
        for f in range(lengths[s]):
            cur = {"K_s0_b44": tt(K[f:f + 1]), "invK_s0_b44": tt(np.linalg.inv(K[f:f + 1])), "cam_T_world_b44": tt(T[f:f + 1]),
                   "world_T_cam_b44": tt(np.linalg.inv(T[f:f + 1]).astype(np.float32)), "K_full_depth_b44": tt(K[f:f + 1]),
                   "_t": gu.to_dev(syn.volume_inputs(1, k, hh, ww, 16, 100 * s + f)),
                   "_pyr": [tt(p).contiguous(memory_format=torch.channels_last)
                            for p in syn.prior_pyramid(1, [64, 64, 128, 256, 512], Hh, Wh, 100 * s + 50 + f)],
                   "_base": tt(surface[f:f + 1] * np.float32(0.6))}
            batches.append((cur, {}))

        def model_fn(cur, src):
            t = cur["_t"]
            out = dict(model.forward_from_features(cur["_pyr"], t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"],
                                                   t["src_Ks"], t["cur_invK"], cur, return_mask=True))
            record.append(out["depth_pred_s0_b1hw"].clone())
            out["depth_pred_s0_b1hw"] = cur["_base"] + 0.02 * torch.tanh(out["depth_pred_s0_b1hw"] - 1.0)
            return out

        fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
        return loops.IncrementalScan(model_fn, fuser, batches, (Hh, Wh)), fuser

    model.enable_launch_programs(programs)
    try:
        alone = []
        for s in range(3):
            rec = []
            scan, fuser = make_scan(s, rec)
            while scan.step():
                pass
            torch.cuda.synchronize()
            t = fuser.tsdf_fuser_pred.tsdf
            alone.append((rec, t.tsdf_values.clone(), t.tsdf_weights.clone(), t.voxel_bitmap.clone()))
            assert scan.frames == lengths[s] and (t.tsdf_weights > 0).sum().item() > 500
        recs = [[], [], []]
        built = [make_scan(s, recs[s]) for s in range(3)]
        done = loops.run_incremental_scans([b[0] for b in built], in_flight=3, device=dev)
        torch.cuda.synchronize()
        assert done == list(lengths)
        for s in range(3):
            t = built[s][1].tsdf_fuser_pred.tsdf
            rec0, v0, w0, a0 = alone[s]
            assert len(recs[s]) == len(rec0) and all(torch.equal(a, b) for a, b in zip(recs[s], rec0)), s
            assert torch.equal(t.tsdf_values.view(torch.int16), v0.view(torch.int16)), s
            assert torch.equal(t.tsdf_weights.view(torch.int16), w0.view(torch.int16)) and torch.equal(t.voxel_bitmap, a0), s
    finally:
        model.enable_launch_programs(False)


@pytest.mark.parametrize("programs", [False, True])
def test_incremental_scans_batched_match_the_same_scans_alone(programs):
    """loops.run_incremental_scans_batched (round 6): three scans of different lengths advanced in lock step, ONE model call
    per turn on the collated frames (batch 3 -> 3 -> 3 -> 2 -> 1 as the scans finish).  The model at batch k sums its convs in
    another order than at batch 1 (other plans), so per scan the results match the scan alone to fp32 rounding, not bit for
    bit: predicted depths within 1e-4 relative (north-star tolerance 1e-3), the final TSDF equal except isolated voxels whose
    fp16 thresholds the rounding crossed (< 0.2 % of the observed voxels differ by more than 0.02 in value or weight)."""
    import gpu_util as gu
    from doubletake_amd import loops
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    hh, ww, k, D2 = 48, 64, 3, 16
    Hh, Wh = 2 * hh, 2 * ww
    model = DepthModelCVHint(4 * hh, 4 * ww, depth_decoder_name="skip", matching_num_depth_bins=D2, model_num_views=k + 1,
                             matching_encoder_type=None)
    gu.set_formula_weights(model, 9)
    model = model.to(dev)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lengths = (5, 3, 4)

    def model_fn(cur, src):
        t = cur["_t"]
        out = dict(model.forward_from_features(cur["_pyr"], t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"],
                                               t["src_Ks"], t["cur_invK"], cur, return_mask=True))
        out["raw_depth"] = out["depth_pred_s0_b1hw"].clone()
        out["depth_pred_s0_b1hw"] = cur["_base"] + 0.02 * torch.tanh(out["depth_pred_s0_b1hw"] - 1.0)
        return out

    def make_scan(s, record):
        surface, K, T = syn.tsdf_frames(lengths[s], Hh, Wh, seed=11 + s, bounds=BD)
        batches = []
        for f in range(lengths[s]):
            cur = {"K_s0_b44": tt(K[f:f + 1]), "invK_s0_b44": tt(np.linalg.inv(K[f:f + 1])), "cam_T_world_b44": tt(T[f:f + 1]),
                   "world_T_cam_b44": tt(np.linalg.inv(T[f:f + 1]).astype(np.float32)), "K_full_depth_b44": tt(K[f:f + 1]),
                   "_t": gu.to_dev(syn.volume_inputs(1, k, hh, ww, 16, 100 * s + f)),
                   "_pyr": [tt(p).contiguous(memory_format=torch.channels_last)
                            for p in syn.prior_pyramid(1, [64, 64, 128, 256, 512], Hh, Wh, 100 * s + 50 + f)],
                   "_base": tt(surface[f:f + 1] * np.float32(0.6))}
            batches.append((cur, {}))
        fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
        scan = loops.IncrementalScan(model_fn, fuser, batches, (Hh, Wh), on_frame=lambda i, c, o: record.append(o["raw_depth"].clone()))
        return scan, fuser

    model.enable_launch_programs(programs)
    try:
        alone = []
        for s in range(3):
            rec = []
            scan, fuser = make_scan(s, rec)
            while scan.step():
                pass
            torch.cuda.synchronize()
            t = fuser.tsdf_fuser_pred.tsdf
            alone.append((rec, t.tsdf_values.float().clone(), t.tsdf_weights.float().clone()))
        recs = [[], [], []]
        built = [make_scan(s, recs[s]) for s in range(3)]
        done = loops.run_incremental_scans_batched([b[0] for b in built], model_fn)
        torch.cuda.synchronize()
        assert done == list(lengths)
        stats = []
        for s in range(3):
            t = built[s][1].tsdf_fuser_pred.tsdf
            rec0, v0, w0 = alone[s]
            assert len(recs[s]) == len(rec0)
            rel = max(float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) for a, b in zip(recs[s], rec0))
            assert all(a.shape == b.shape == (1, 1, Hh, Wh) for a, b in zip(recs[s], rec0)) and rel <= 1e-4, (s, rel)
            seen = (w0 > 0) | (t.tsdf_weights > 0)
            off = seen & (((t.tsdf_values.float() - v0).abs() > 0.02) | ((t.tsdf_weights.float() - w0).abs() > 0.02))
            frac = off.sum().item() / max(1, seen.sum().item())
            stats.append((rel, frac, int(seen.sum())))
            assert seen.sum().item() > 500 and frac < 2e-3, (s, frac)
        print("batched vs alone (max rel depth diff, differing voxel fraction, observed voxels):", stats)
    finally:
        model.enable_launch_programs(False)
