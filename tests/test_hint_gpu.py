"""GPU: hint-mesh depth rasteriser vs the hand-derived PyTorch3D known-answer cases and the numpy oracle, and the closed incremental hint loop
fuse -> marching cubes -> render -> sample -> volume -> decoder -> fuse (reference test_incremental.py:187-372)."""
import numpy as np
import pytest
import torch

from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

BD = dict(xmin=-1.28, xmax=1.28, ymin=-1.12, ymax=1.12, zmin=0.0, zmax=2.24)


def _fused(nframes=3, H=120, W=160):
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser

    depth, K, T = syn.tsdf_frames(5, H, W, seed=3, bounds=BD)
    depth = depth * np.float32(0.6)
    fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    for f in range(nframes):
        fuser.fuse_frames(*(torch.from_numpy(a[f:f + 1]).to(gu.dev()) for a in (depth, K, T)), None)
    return fuser, depth, K, T


def _handcases():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "raster_handcases.json")))


@pytest.mark.parametrize("case", _handcases(), ids=lambda c: c["name"])
def test_raster_reproduces_hand_derived_pytorch3d_cases(case):
    """dt_raster_depth_f32 (through the reference-shaped MeshDepthRenderer, i.e. with normalised intrinsics) against the
    known answers derived from PyTorch3D 0.7.4's rules (tests/golden/make_raster_handcases.py): camera conversion, pixel
    grid of non-square images, STRICT coverage on edges, perspective-correct depth, nearest face, background -1."""
    import gpu_util as gu
    from doubletake_amd.tools.tsdf import Meshes
    from doubletake_amd.utils.rendering_utils import MeshDepthRenderer

    h, w = case["h"], case["w"]
    want = np.asarray(case["expected"], dtype=np.float64)
    verts = torch.tensor(case["verts"], dtype=torch.float32, device=gu.dev())
    faces = torch.tensor(case["faces"], dtype=torch.int64, device=gu.dev())
    K = torch.tensor(case["K"], dtype=torch.float32, device=gu.dev())[None].clone()
    K[:, 0] /= w
    K[:, 1] /= h
    T = torch.tensor(case["cam_T_world"], dtype=torch.float32, device=gu.dev())[None]
    got, _ = MeshDepthRenderer(h, w).render(Meshes(verts=[verts], faces=[faces]), T, K)
    got = got[0, 0].cpu().numpy()
    np.testing.assert_array_equal(got > 0, want > 0)
    np.testing.assert_allclose(got[want > 0], want[want > 0], rtol=2e-6)
    assert np.all(got[want < 0] == -1)


def test_raster_vs_oracle_on_fused_mesh():
    import gpu_util as gu
    from doubletake_amd.utils.rendering_utils import MeshDepthRenderer
    from oracle import raster_ref

    fuser, depth, K, T = _fused()
    mesh, verts, faces = fuser.get_mesh_pytorch3d()
    assert faces.shape[0] > 2000
    h, w = 60, 80
    Kh = K[1].copy()
    Kh[:2] *= 0.5  # intrinsics of the half-resolution render
    Kn = torch.from_numpy(Kh[None]).to(gu.dev()).clone()
    Kn[:, 0] /= w
    Kn[:, 1] /= h
    r = MeshDepthRenderer(h, w)
    got, _ = r.render(mesh, torch.from_numpy(T[1:2]).to(gu.dev()), Kn)
    got = got[0, 0].cpu().numpy()
    want = raster_ref.render_depth(verts.cpu().numpy(), faces.cpu().numpy(), T[1], Kh, h, w)
    both = (got > 0) & (want > 0)
    assert both.mean() > 0.3
    # coverage may differ on a handful of edge pixels (fp32 vs fp64 edge functions)
    assert ((got > 0) != (want > 0)).mean() < 0.005
    assert np.abs(got[both] - want[both]).max() < 2e-3
    assert np.median(np.abs(got[both] - want[both])) < 1e-5


def test_render_reproduces_fused_depth():
    """Self-consistency of fuse -> marching cubes -> raster: rendering the TSDF mesh from a fused
    camera gives back that frame's depth to within ~1.5 voxels (0.04 m) on most pixels."""
    import gpu_util as gu
    from doubletake_amd.utils.rendering_utils import MeshDepthRenderer

    fuser, depth, K, T = _fused(nframes=1)
    mesh, _, _ = fuser.get_mesh_pytorch3d()
    H, W = depth.shape[-2:]
    Kn = torch.from_numpy(K[0:1]).to(gu.dev()).clone()
    Kn[:, 0] /= W
    Kn[:, 1] /= H
    got, _ = MeshDepthRenderer(H, W).render(mesh, torch.from_numpy(T[0:1]).to(gu.dev()), Kn)
    got = got[0, 0].cpu().numpy()
    valid = got > 0
    assert valid.mean() > 0.5
    err = np.abs(got - depth[0, 0])[valid]
    assert np.median(err) < 0.02 and np.quantile(err, 0.9) < 0.06


def test_incremental_hint_loop_end_to_end():
    import gpu_util as gu
    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.utils.rendering_utils import MeshDepthRenderer, empty_hint, prepare_mesh_hint

    dev = gu.dev()
    h, w, k, D = 32, 40, 2, 16
    H2, W2 = 2 * h, 2 * w
    model = DepthModelCVHint(4 * h, 4 * w, depth_decoder_name="skip", matching_num_depth_bins=D, model_num_views=k + 1)
    gu.set_formula_weights(model, 7)
    model = model.to(dev)
    _, K, T = syn.tsdf_frames(3, H2, W2, seed=3, bounds=BD)
    fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    renderer = MeshDepthRenderer(H2, W2)
    coverage = []
    for f in range(3):
        inp = syn.volume_inputs(1, k, h, w, 16, 10 + f)
        t = gu.to_dev(inp)
        pyr = [torch.from_numpy(p).to(dev) for p in syn.prior_pyramid(1, [64, 64, 128, 256, 512], H2, W2, 20 + f)]
        # static camera: the three keyframes see the same surface, so weights accumulate
        cur = {"K_s0_b44": torch.from_numpy(K[0:1]).to(dev), "invK_s0_b44": torch.from_numpy(np.linalg.inv(K[0:1])).to(dev),
               "cam_T_world_b44": torch.from_numpy(T[0:1]).to(dev),
               "world_T_cam_b44": torch.from_numpy(np.linalg.inv(T[0:1])).float().to(dev)}
        if f == 0:
            empty_hint(cur, torch.zeros(1, 1, H2, W2, device=dev))
        else:
            prepare_mesh_hint(fuser, renderer, cur, H2, W2)
        coverage.append(cur["depth_hint_mask_b1hw"].mean().item())
        out = model.forward_from_features(pyr, t["cur_feats"], t["src_feats"], t["src_extrinsics"], t["src_poses"], t["src_Ks"],
                                          t["cur_invK"], cur, return_mask=True)
        d0 = out["depth_pred_s0_b1hw"]
        assert torch.isfinite(d0).all() and tuple(d0.shape) == (1, 1, H2, W2)
        # fuse a plausible surface (random-weight networks do not predict metric depth)
        fuser.fuse_frames(d0.clamp(0.9, 1.2), cur["K_s0_b44"], cur["cam_T_world_b44"], None)
        # hint maps obey the reference invariants: NaN exactly where mask == 0, weights zero there
        hm = cur["depth_hint_mask_b_b1hw"]
        assert torch.equal(torch.isnan(cur["depth_hint_b1hw"]), ~hm)
        assert (cur["sampled_weights_b1hw"][~hm] == 0).all()
        if f > 0:
            assert (cur["sampled_weights_b1hw"][hm] >= 0.025).all()
    # one observation gives weights <= 2.5/100 * conf < 0.025 (tools/tsdf.py:546-549): the reference's
    # 0.025 cut masks everything until a voxel has been seen twice
    assert coverage[0] == 0.0 and coverage[1] == 0.0 and coverage[2] > 0.2


def test_fused_hint_preparation_matches_composed_version():
    """prepare_mesh_hint_fused (soup render + one back-project/sample/threshold kernel) against
    prepare_mesh_hint (the reference's sequence of torch ops over the merged mesh)."""
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.utils.rendering_utils import MeshDepthRenderer, prepare_mesh_hint, prepare_mesh_hint_fused

    dev = gu.dev()
    H2, W2 = 96, 128
    depth, K, T = syn.tsdf_frames(4, H2, W2, seed=5, bounds=BD)
    depth = depth * np.float32(0.6)
    fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
    for _ in range(3):  # the same views several times so that weights pass the 0.025 cut
        fuser.fuse_frames(d, k, t, None)
    for j in (0, 2):
        mk = lambda: {"K_s0_b44": k[j:j + 1], "invK_s0_b44": torch.linalg.inv(k[j:j + 1]), "cam_T_world_b44": t[j:j + 1],
                      "world_T_cam_b44": torch.linalg.inv(t[j:j + 1])}
        a, b = mk(), mk()
        da = prepare_mesh_hint(fuser, MeshDepthRenderer(H2, W2), a, H2, W2)
        db = prepare_mesh_hint_fused(fuser, b, H2, W2)
        covered = (da > 0) & (db > 0)
        assert covered.float().mean() > 0.3
        assert ((da > 0) != (db > 0)).float().mean() < 2e-3           # edge pixels only
        assert (da - db)[covered].abs().max() < 1e-4
        ma, mb = a["depth_hint_mask_b_b1hw"], b["depth_hint_mask_b_b1hw"]
        assert mb.dtype == torch.bool and ma.float().mean() > 0.2
        assert (ma != mb).float().mean() < 2e-3
        both = ma & mb
        assert (a["sampled_weights_b1hw"] - b["sampled_weights_b1hw"])[both].abs().max() < 1e-4
        assert (a["depth_hint_b1hw"] - b["depth_hint_b1hw"])[both].abs().max() < 1e-4
        assert torch.equal(torch.isnan(b["depth_hint_b1hw"]), ~mb)
        assert torch.equal(b["depth_hint_mask_b1hw"], mb.float())
        assert (b["sampled_weights_b1hw"][~mb] == 0).all()


def test_fused_marching_cubes_render_is_bitwise_the_soup_render():
    """dt_mc_raster_depth_f32 (one kernel, no vertex buffer, no host read) against count -> generate -> soup raster:
    same triangles, same vertex arithmetic, order-independent depth test -> identical bits, and nothing is read back."""
    import gpu_util as gu
    from doubletake_amd import _abi
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.utils.rendering_utils import prepare_mesh_hint_fused

    dev = gu.dev()
    coverage = []
    for (H2, W2, vox, scale) in ((96, 128, 0.04, 0.6), (240, 320, 0.04, 0.6), (120, 160, 0.08, 0.6)):
        depth, K, T = syn.tsdf_frames(5, H2, W2, seed=7, bounds=BD)
        depth = depth * np.float32(scale)
        fuser = OurFuser(None, vox, 3.0, bounds=BD)
        d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
        for _ in range(3):
            fuser.fuse_frames(d, k, t, None)
        for j in (0, 3):
            mk = lambda: {"K_s0_b44": k[j:j + 1], "invK_s0_b44": torch.linalg.inv(k[j:j + 1]), "cam_T_world_b44": t[j:j + 1],
                          "world_T_cam_b44": torch.linalg.inv(t[j:j + 1])}
            a, b = mk(), mk()
            da = prepare_mesh_hint_fused(fuser, a, H2, W2, via_soup=True)
            torch.cuda.synchronize()
            n0 = _abi.lib().dt_kernel_launch_count()
            db = prepare_mesh_hint_fused(fuser, b, H2, W2)
            assert _abi.lib().dt_kernel_launch_count() - n0 == 4   # z-buffer init, mc+raster, resolve, hint maps
            torch.cuda.synchronize()
            coverage.append(float((da > 0).float().mean()))
            assert torch.equal(da.view(torch.int32), db.view(torch.int32))
            for key in ("depth_hint_b1hw", "sampled_weights_b1hw", "depth_hint_mask_b1hw"):
                assert torch.equal(a[key].view(torch.int32), b[key].view(torch.int32)), key
    assert max(coverage) > 0.3 and sum(c > 0.1 for c in coverage) >= 4, coverage
    # an empty volume renders the background
    fuser = OurFuser(None, 0.04, 3.0, bounds=BD)
    c = {"K_s0_b44": k[:1], "invK_s0_b44": torch.linalg.inv(k[:1]), "cam_T_world_b44": t[:1], "world_T_cam_b44": torch.linalg.inv(t[:1])}
    assert (prepare_mesh_hint_fused(fuser, c, 120, 160) == -1).all()


def test_fused_marching_cubes_render_on_a_noisy_volume_with_bounds():
    """Stress: random TSDF values (up to five triangles in most cells, every workgroup's LDS compaction filled), a random
    half of the voxels active, a sub-box given as min / max bounds -- fused kernel against count -> generate -> soup raster,
    bit for bit, through the C ABI."""
    import ctypes as C

    import gpu_util as gu
    from doubletake_amd import _abi
    from doubletake_amd.utils.pytorch3d_extras import marching_cubes_raw

    dev = gu.dev()
    L = _abi.lib()
    X, Y, Z = 40, 48, 32
    vs, h, w = 0.05, 120, 160
    stream = _abi.current_stream(dev)
    vol = torch.from_numpy(syn.hash_u01((X, Y, Z), 91) * 1.6 - 0.8).to(dev).half().contiguous()
    active_mask = torch.from_numpy(syn.hash_u01((X * Y * Z,), 92) > 0.5).to(dev)
    words = active_mask.view(-1, 32).to(torch.int64)
    words = (words << torch.arange(32, device=dev, dtype=torch.int64)).sum(1)   # bit i of a word = voxel 32 * word + i
    bitmap = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()
    o = (C.c_float * 3)(-1.0, -1.2, 0.5)
    K = torch.tensor([[150.0, 0, 80, 0], [0, 150.0, 60, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev)
    T = torch.eye(4, device=dev)
    T[:3, 3] = torch.tensor([0.0, 0.0, 0.6])
    for mn, mx in ((None, None), ((5, 6, 3), (30, 40, 28))):
        verts, _, _ = marching_cubes_raw(vol, bitmap, 0.0, mn, mx)
        nf = int(verts.shape[0]) // 3
        assert nf > 20000
        ws = torch.empty(h * w, device=dev, dtype=torch.int32)
        want = torch.empty(h, w, device=dev)
        _abi.check(L.dt_raster_soup_depth_f32(_abi.ptr(verts), nf, o, vs, _abi.ptr(T), _abi.ptr(K), h, w, _abi.ptr(ws),
                                              _abi.ptr(want), stream), "soup")
        got = torch.empty(h, w, device=dev)
        ib = lambda b: None if b is None else (C.c_int * 3)(*b)
        _abi.check(L.dt_mc_raster_depth_f32(_abi.ptr(vol), _abi.ptr(bitmap), X, Y, Z, 0.0, ib(mn), ib(mx), o, vs, _abi.ptr(T),
                                            _abi.ptr(K), h, w, _abi.ptr(ws), _abi.ptr(got), stream), "fused")
        torch.cuda.synchronize()
        assert (want > 0).float().mean() > 0.5
        assert torch.equal(want.view(torch.int32), got.view(torch.int32))
