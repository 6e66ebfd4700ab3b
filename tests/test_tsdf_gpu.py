"""GPU parity of the fp16 TSDF fuser / sampler and the active-voxel marching cubes."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from doubletake_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

BD = dict(xmin=-1.28, xmax=1.28, ymin=-1.12, ymax=1.12, zmin=0.0, zmax=2.24)
RUNS = {"a": (0.04, 3.0, False, 120, 160, 3), "b": (0.04, 3.0, True, 96, 128, 4)}


def _run(tag, nframes=5):
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser
    from oracle import tsdf_ref as tr

    vs, maxd, ext, H, W, seed = RUNS[tag]
    depth, K, T = syn.tsdf_frames(5, H, W, seed=seed, bounds=BD)
    depth = depth * np.float32(0.6)
    fuser = OurFuser(gt_path=None, fusion_resolution=vs, max_fusion_depth=maxd, extended_neg_truncation=ext, bounds=BD)
    vol = tr.TSDFVolume(BD, vs)
    undefined = np.zeros(vol.values.size, dtype=bool)
    snaps = {}
    for f in range(nframes):
        fuser.fuse_frames(torch.from_numpy(depth[f:f + 1]).to(gu.dev()), torch.from_numpy(K[f:f + 1]).to(gu.dev()),
                          torch.from_numpy(T[f:f + 1]).to(gu.dev()), None)
        tr.integrate(vol, depth[f, 0], K[f], T[f], maxd, extended_neg_truncation=ext)
        undefined[vol.last_undefined_ids] = True
        t = fuser.tsdf_fuser_pred.tsdf
        snaps[f + 1] = (t.tsdf_values.cpu().numpy().copy(), t.tsdf_weights.cpu().numpy().copy(),
                        t.active_keys().cpu().numpy().astype(np.int64), vol.values.copy(), vol.weights.copy(),
                        np.array(sorted(vol.active), dtype=np.int64).reshape(-1, 3), undefined.copy())
    return fuser, vol, snaps


@pytest.mark.parametrize("tag", ["a", "b"])
def test_integrate_bit_exact_vs_oracle_and_reference_golden(tag):
    g = load_golden("tsdf.npz")
    fuser, vol, snaps = _run(tag)
    assert list(fuser.tsdf_fuser_pred.tsdf.tsdf_values.shape) == list(g[f"int_{tag}_dims"])
    for n in (1, 2, 5):
        gv, gw, gk, ov, ow, ok_, undef = snaps[n]
        # vs the numpy oracle: every voxel, bit for bit
        np.testing.assert_array_equal(gv.view(np.uint16), ov.view(np.uint16))
        np.testing.assert_array_equal(gw.view(np.uint16), ow.view(np.uint16))
        np.testing.assert_array_equal(gk, ok_)
        # vs the reference's own CPU-half run (golden), except where its sampler hits C++ UB
        keep = ~undef
        np.testing.assert_array_equal(gv.reshape(-1)[keep].view(np.uint16), g[f"int_{tag}_vals_{n}"].reshape(-1)[keep].view(np.uint16))
        np.testing.assert_array_equal(gw.reshape(-1)[keep].view(np.uint16), g[f"int_{tag}_wts_{n}"].reshape(-1)[keep].view(np.uint16))
        np.testing.assert_array_equal(gk, g[f"int_{tag}_active_{n}"])  # the integer active-key set


@pytest.mark.parametrize("tag", ["a", "b"])
def test_sample_tsdf_vs_reference_golden(tag):
    import gpu_util as gu

    g = load_golden("tsdf.npz")
    fuser, _, _ = _run(tag)
    t = fuser.tsdf_fuser_pred.tsdf
    # sample the golden volume itself so the comparison is independent of the UB voxels
    t.tsdf_values.copy_(torch.from_numpy(g[f"int_{tag}_vals_5"]).to(gu.dev()))
    t.tsdf_weights.copy_(torch.from_numpy(g[f"int_{tag}_wts_5"]).to(gu.dev()))
    pts = torch.from_numpy(g[f"sample_{tag}_pts"]).to(gu.dev())
    np.testing.assert_allclose(fuser.sample_tsdf(pts, "weights").cpu().numpy(), g[f"sample_{tag}_weights"], atol=1e-6)
    np.testing.assert_allclose(fuser.sample_tsdf(pts, "tsdf").cpu().numpy(), g[f"sample_{tag}_tsdf"], atol=1e-6)


def test_batched_integrate_equals_sequential():
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser

    vs, maxd, ext, H, W, seed = RUNS["a"]
    depth, K, T = syn.tsdf_frames(4, H, W, seed=seed, bounds=BD)
    depth = depth * np.float32(0.6)
    d, k, t = (torch.from_numpy(a).to(gu.dev()) for a in (depth, K, T))
    f1 = OurFuser(None, vs, maxd, bounds=BD)
    f2 = OurFuser(None, vs, maxd, bounds=BD)
    f1.fuse_frames(d, k, t, None)
    for i in range(4):
        f2.fuse_frames(d[i:i + 1], k[i:i + 1], t[i:i + 1], None)
    a, b = f1.tsdf_fuser_pred.tsdf, f2.tsdf_fuser_pred.tsdf
    assert torch.equal(a.tsdf_values, b.tsdf_values) and torch.equal(a.tsdf_weights, b.tsdf_weights)
    assert torch.equal(a.voxel_bitmap, b.voxel_bitmap)


def test_marching_cubes_vs_oracle_on_fused_volume():
    from doubletake_amd.utils.pytorch3d_extras import marching_cubes_raw
    from oracle import mc_ref

    fuser, vol, snaps = _run("a", nframes=3)
    t = fuser.tsdf_fuser_pred.tsdf
    keys = t.active_keys().cpu().numpy().astype(np.int64)
    v, f, ids = marching_cubes_raw(t.tsdf_values, t.voxel_bitmap, 0.0)
    ov, of, oids = mc_ref.marching_cubes_active(vol.values.astype(np.float32), keys, 0.0)
    assert len(f) == len(of) > 500
    # same order too (ascending voxel id), so compare directly
    np.testing.assert_allclose(v.cpu().numpy(), ov, atol=1e-5)
    np.testing.assert_array_equal(ids.cpu().numpy(), oids)
    np.testing.assert_array_equal(f.cpu().numpy(), of)
    # reference wrapper post-processing (dedup by edge id, axis flip, world scale)
    mesh, wv, wf = fuser.get_mesh_pytorch3d()
    pv, pf = mc_ref.postprocess(ov, of, oids)
    want_v = t.origin.float().numpy().reshape(1, 3) + pv * np.float32(t.voxel_size)
    np.testing.assert_allclose(wv.cpu().numpy(), want_v, atol=1e-4)
    np.testing.assert_array_equal(wf.cpu().numpy(), pf)
    assert mesh.verts_list()[0].shape[1] == 3
    # bounds argument restricts cells (CUDA semantics: min <= g < max)
    mn, mx = [0, 0, 0], [30, 56, 56]
    v2, f2, _ = marching_cubes_raw(t.tsdf_values, t.voxel_bitmap, 0.0, mn, mx)
    ov2, of2, _ = mc_ref.marching_cubes_active(vol.values.astype(np.float32), keys, 0.0, mn, mx)
    assert len(f2) == len(of2) and 0 < len(f2) < len(f)


def test_marching_cubes_matches_reference_cpu_semantics_where_they_coincide():
    """Dense active set, no unobserved corners: triangle set equals the (golden-pinned) oracle's."""
    import gpu_util as gu
    from doubletake_amd.utils.pytorch3d_extras import keys_to_bitmap, marching_cubes_raw
    from oracle import mc_ref

    n = 16
    g = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float32)] * 3, indexing="ij"), -1)
    vol = (np.linalg.norm(g - np.array([7.3, 7.71, 6.9], dtype=np.float32), axis=-1) - np.float32(4.37)) * np.float32(0.2)
    vol = np.clip(vol, -0.9, 0.9).astype(np.float16)
    keys = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3)
    tv = torch.from_numpy(vol).to(gu.dev())
    bm = keys_to_bitmap(torch.from_numpy(keys).to(gu.dev()), (n, n, n))
    v, f, ids = marching_cubes_raw(tv, bm, 0.0)
    ov, of, _ = mc_ref.marching_cubes_active(vol.astype(np.float32), keys, 0.0)
    assert mc_ref.triangle_set(v.cpu().numpy(), f.cpu().numpy()) == mc_ref.triangle_set(ov, of)


def test_full_size_volume_one_frame_vs_oracle():
    """BASELINE-size fuser: 8 x 8 x 3.2 m at 0.04 m (200x200x80), 480x640 frame, vs the oracle bit for bit."""
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser
    from oracle import tsdf_ref as tr

    bd = dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2)
    depth, K, T = syn.tsdf_frames(2, 480, 640, seed=9, bounds=bd)
    fuser = OurFuser(None, 0.04, 3.0, bounds=bd)
    vol = tr.TSDFVolume(bd, 0.04)
    for f in range(2):
        fuser.fuse_frames(*(torch.from_numpy(a[f:f + 1]).to(gu.dev()) for a in (depth, K, T)), None)
        tr.integrate(vol, depth[f, 0], K[f], T[f], 3.0)
    t = fuser.tsdf_fuser_pred.tsdf
    np.testing.assert_array_equal(t.tsdf_values.cpu().numpy().view(np.uint16), vol.values.view(np.uint16))
    np.testing.assert_array_equal(t.tsdf_weights.cpu().numpy().view(np.uint16), vol.weights.view(np.uint16))
    np.testing.assert_array_equal(t.active_keys().cpu().numpy().astype(np.int64),
                                  np.array(sorted(vol.active), dtype=np.int64).reshape(-1, 3))
    assert len(vol.active) > 10000


def test_batched_integrate_equals_frame_by_frame():
    """One launch over 5 frames (ordered per voxel, value kept in registers) == 5 single-frame launches, bit for bit."""
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser

    vs, maxd, ext, H, W, seed = RUNS["a"]
    depth, K, T = syn.tsdf_frames(5, H, W, seed=seed, bounds=BD)
    depth = depth * np.float32(0.6)
    d, k, t = (torch.from_numpy(a).to(gu.dev()) for a in (depth, K, T))
    one = OurFuser(None, vs, maxd, bounds=BD)
    for f in range(5):
        one.fuse_frames(d[f:f + 1], k[f:f + 1], t[f:f + 1], None)
    allf = OurFuser(None, vs, maxd, bounds=BD)
    allf.fuse_frames(d, k, t, None)
    a, b = one.tsdf_fuser_pred.tsdf, allf.tsdf_fuser_pred.tsdf
    assert torch.equal(a.tsdf_values.view(torch.int16), b.tsdf_values.view(torch.int16))
    assert torch.equal(a.tsdf_weights.view(torch.int16), b.tsdf_weights.view(torch.int16))
    assert torch.equal(a.voxel_bitmap, b.voxel_bitmap)
    assert (b.tsdf_weights > 0).sum().item() > 1000


def test_host_level_helpers_match_reference_semantics(tmp_path):
    """get_frustum_bounds / project_to_camera / fuser properties / save_mesh (tools/tsdf.py:15-50,257-265,373-412)."""
    import gpu_util as gu
    from doubletake_amd.tools.fusers_helper import OurFuser
    from doubletake_amd.tools.tsdf import get_frustum_bounds
    from doubletake_amd.utils.formats import read_ply

    vs, maxd, ext, H, W, seed = RUNS["a"]
    depth, K, T = syn.tsdf_frames(2, H, W, seed=seed, bounds=BD)
    depth = depth * np.float32(0.6)
    d, k, t = (torch.from_numpy(a).to(gu.dev()) for a in (depth, K, T))
    fuser = OurFuser(None, vs, maxd, bounds=BD)
    fuser.fuse_frames(d, k, t, None)
    tf = fuser.tsdf_fuser_pred
    # frustum box: every corner of the near/far planes lies inside it and it is tight
    invK, pose = torch.linalg.inv(k[0:1]), torch.linalg.inv(t[0:1])
    lo, hi = get_frustum_bounds(invK, pose, 0.5, 3.0, H, W)
    uv = torch.tensor([[0, 0, 1, 1], [W, 0, 1, 1], [0, H, 1, 1], [W, H, 1, 1]], dtype=torch.float32, device=gu.dev()).t()
    pts = []
    for dd in (0.5, 3.0):
        c = invK[0] @ uv
        c[:3] *= dd
        pts.append((pose[0] @ c)[:3])
    pts = torch.cat(pts, 1)
    assert torch.allclose(lo, pts.amin(1), atol=1e-5) and torch.allclose(hi, pts.amax(1), atol=1e-5)
    # project_to_camera: the principal ray at depth 2 m lands on the principal point with z = 2
    centre = torch.tensor([[k[0, 0, 2]], [k[0, 1, 2]], [1.0], [1.0]], device=gu.dev())
    X = invK[0] @ centre
    X[:3] *= 2.0
    world = (pose[0] @ X).unsqueeze(0)
    cam = tf.project_to_camera(t[0:1], k[0:1], world)
    assert torch.allclose(cam[0, :, 0], torch.stack([k[0, 0, 2], k[0, 1, 2], torch.tensor(2.0, device=gu.dev())]), atol=1e-3)
    assert abs(tf.truncation - 3 * vs) < 1e-9 and tuple(tf.voxel_coords_3hwd.shape[1:]) == tuple(tf.shape)
    keys = tf.voxel_hashset
    assert keys.shape[1] == 3 and keys.shape[0] > 100
    fuser.tsdf_fuser_pred.tsdf.save_mesh(str(tmp_path), "scene.bin")
    v, f = read_ply(str(tmp_path / "scene.ply"))
    assert v.shape[0] > 100 and f.max() < v.shape[0]


def test_xslab_integration_assembles_the_whole_volume_bit_for_bit():
    """dt_tsdf_integrate_frames_xslab_f16 (multi-GPU voxel-slab fusion) -- three 'ranks' emulated on
    one GPU: each integrates every frame into its x-slab of its own volume; stitched together, values / weights / active
    bits equal the volume integrated whole (voxel centres come from the global index)."""
    import gpu_util as gu
    from doubletake_amd import parallel as par
    from doubletake_amd.tools.fusers_helper import OurFuser

    dev = gu.dev()
    bd = dict(xmin=-1.28, xmax=1.28, ymin=-1.12, ymax=1.12, zmin=0.0, zmax=2.24)
    H, W = 120, 160
    depth, K, T = syn.tsdf_frames(6, H, W, seed=3, bounds=bd)
    depth = depth * np.float32(0.6)
    d, k, t = (torch.from_numpy(a).to(dev) for a in (depth, K, T))
    whole = OurFuser(None, 0.04, 3.0, bounds=bd)
    for sl in (slice(0, 2), slice(2, 5), slice(5, 6)):
        whole.fuse_frames(d[sl], k[sl], t[sl], None)
    ref = whole.tsdf_fuser_pred.tsdf
    X = ref.tsdf_values.shape[0]
    world = 3
    parts = []
    for r in range(world):
        f = OurFuser(None, 0.04, 3.0, bounds=bd)
        f.tsdf_fuser_pred.x_range = par.slab_bounds(X, world, r)
        for sl in (slice(0, 2), slice(2, 5), slice(5, 6)):
            f.fuse_frames(d[sl], k[sl], t[sl], None)
        parts.append(f.tsdf_fuser_pred.tsdf)
    torch.cuda.synchronize()
    assert (ref.tsdf_weights > 0).sum().item() > 10000
    for r, p in enumerate(parts):
        x0, x1 = par.slab_bounds(X, world, r)
        assert (ref.tsdf_weights[x0:x1] > 0).any()  # every slab sees geometry: the comparison is not vacuous
        assert torch.equal(p.tsdf_values[x0:x1].view(torch.int16), ref.tsdf_values[x0:x1].view(torch.int16))
        assert torch.equal(p.tsdf_weights[x0:x1].view(torch.int16), ref.tsdf_weights[x0:x1].view(torch.int16))
        assert torch.equal(p.voxel_bitmap.view(X, -1)[x0:x1], ref.voxel_bitmap.view(X, -1)[x0:x1])
        # nothing outside the slab was touched
        fresh = OurFuser(None, 0.04, 3.0, bounds=bd).tsdf_fuser_pred.tsdf
        for a, b in ((p.tsdf_values, fresh.tsdf_values), (p.tsdf_weights, fresh.tsdf_weights)):
            assert torch.equal(a[:x0], b[:x0]) and torch.equal(a[x1:], b[x1:])
