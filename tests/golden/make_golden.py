#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (nianticlabs/doubletake) in THIS container.

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (read-only) -- it does NOT exist on the GPU box, which is why the
outputs are committed as small .npz fixtures next to this script.

How the reference is imported: its hot-path modules pull in
kornia / torchvision / timm / antialiased_cnns / open3d / pytorch3d / skimage / trimesh at
import time, none of which are installed here and none of which are touched by the functions
we call.  We put empty stand-in *packages* for those names on sys.path (written to a temp
dir at run time; only what import-time code needs), then run the reference's own classes on
CPU with closed-form inputs from doubletake_amd.utils.synthetic and formula weights.

Only data (inputs are regenerated from seeds; expected outputs are stored) is written.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF_SRC = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def make_stubs(root):
    w = lambda rel, text="": _write(os.path.join(root, rel), text)
    w("kornia/__init__.py", "from . import filters\n")
    w(
        "kornia/filters.py",
        "import torch\nfrom typing import Tuple\n"
        "def blur_pool2d(x: torch.Tensor, kernel_size: int) -> torch.Tensor:\n    return x\n"
        "def gaussian_blur2d(x: torch.Tensor, k: Tuple[int, int], s: Tuple[float, float]) -> torch.Tensor:\n    return x\n"
        "def spatial_gradient(x: torch.Tensor) -> torch.Tensor:\n    return x\n",
    )
    w("torchvision/__init__.py", "from . import transforms, models, ops\n")
    w("torchvision/transforms/__init__.py", "from . import functional\n")
    w("torchvision/transforms/functional.py")
    w("torchvision/models/__init__.py")
    w("torchvision/ops/__init__.py", "class FeaturePyramidNetwork:\n    pass\n")
    w("antialiased_cnns/__init__.py")
    w("timm/__init__.py")
    # TSDF side
    w(
        "open3d/__init__.py",
        "from . import core\n",
    )
    w(
        "open3d/core.py",
        "import torch\nint64 = 'int64'\n"
        "class Device:\n    def __init__(self, *a):\n        pass\n"
        "class _T:\n    def __init__(self, t):\n        self.t = t\n        self.shape = tuple(t.shape)\n"
        "class Tensor:\n    @staticmethod\n    def from_dlpack(cap):\n        return _T(torch.utils.dlpack.from_dlpack(cap))\n"
        "class HashSet:\n"
        "    def __init__(self, *a, **k):\n        self.keys = set()\n"
        "    def insert(self, t):\n        self.keys.update(map(tuple, t.t.tolist()))\n"
        "class cuda:\n    @staticmethod\n    def release_cache():\n        pass\n",
    )
    for p in ("structures", "renderer", "transforms", "utils"):
        w(f"pytorch3d/{p}/__init__.py", "class Meshes: pass\nclass TexturesVertex: pass\nclass Translate: pass\n")
    w("pytorch3d/__init__.py")
    w("skimage/__init__.py", "from . import measure\n")
    w("skimage/measure.py")
    w("trimesh/__init__.py", "class Trimesh: pass\n")


def import_reference():
    stubs = tempfile.mkdtemp(prefix="dt_stubs_")
    make_stubs(stubs)
    sys.path[:0] = [stubs, REF_SRC]
    import torch  # noqa

    torch.set_num_threads(8)
    import doubletake.utils  # noqa

    fake = types.ModuleType("doubletake.utils.pytorch3d_extras")
    fake.marching_cubes = None
    sys.modules["doubletake.utils.pytorch3d_extras"] = fake
    with contextlib.redirect_stdout(io.StringIO()):
        from doubletake.modules import cost_volume, feature_volume, mesh_hint_volume, networks, networks_fast, layers
        from doubletake.tools import tsdf
        from doubletake.utils import geometry_utils
    return dict(
        cost_volume=cost_volume,
        feature_volume=feature_volume,
        mesh_hint_volume=mesh_hint_volume,
        networks=networks,
        networks_fast=networks_fast,
        layers=layers,
        tsdf=tsdf,
        geometry_utils=geometry_utils,
    )


def set_formula_weights(module, seed, scale_mult=1.0):
    """Overwrite every parameter of a torch module with doubletake_amd.utils.synthetic.formula_weights.

    Parameters are visited in state_dict order; parameter j gets seed + 1000*j.  Weights use
    U(-a, a) with a = scale_mult * sqrt(3 / fan_in) (variance-preserving), biases U(-0.1, 0.1).
    The same rule is re-applied by the tests to build identical weights without torch RNG.
    """
    import torch
    from doubletake_amd.utils.synthetic import formula_weights

    with torch.no_grad():
        for j, (name, p) in enumerate(module.named_parameters()):
            shp = tuple(p.shape)
            if p.ndim > 1:
                fan_in = int(np.prod(shp[1:]))
                a = scale_mult * np.sqrt(3.0 / fan_in)
            else:
                a = 0.1
            p.copy_(torch.from_numpy(formula_weights(shp, seed + 1000 * j, scale=a)))


def t(x):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x))


def checksum(a: np.ndarray, nprobe=64):
    a = np.asarray(a, dtype=np.float32)
    flat = a.reshape(-1)
    idx = (np.arange(nprobe, dtype=np.int64) * 2654435761 % flat.size).astype(np.int64)
    return dict(
        sum=np.float64(flat.astype(np.float64).sum()),
        abssum=np.float64(np.abs(flat.astype(np.float64)).sum()),
        min=np.float32(flat.min()),
        max=np.float32(flat.max()),
        probe_idx=idx,
        probe_val=flat[idx].copy(),
    )


def gen_volume(ref, out):
    import torch
    from doubletake_amd.utils import synthetic as syn

    cases = {
        # name: (b, k, h, w, D, seed, empty_hint, behind_view)
        "k2_land": (1, 2, 24, 32, 8, 1, False, False),
        "k7_land": (1, 7, 24, 32, 8, 2, False, True),
        "k7_b2": (2, 7, 24, 32, 8, 3, False, False),
        "k3_portrait": (1, 3, 32, 24, 8, 4, False, False),
        "k7_empty": (1, 7, 24, 32, 8, 5, True, False),
        "k2_ragged": (1, 2, 19, 27, 5, 6, False, True),
    }
    for name, (b, k, h, w, D, seed, empty, behind) in cases.items():
        inp = syn.volume_inputs(b, k, h, w, 16, seed, empty_hint=empty, behind_view=behind)
        ti = {n: t(v) for n, v in inp.items()}
        hint = {n: ti[n] for n in ("depth_hint_b1hw", "depth_hint_mask_b1hw", "sampled_weights_b1hw")}
        common = dict(
            cur_feats=ti["cur_feats"],
            src_feats=ti["src_feats"],
            src_extrinsics=ti["src_extrinsics"],
            src_poses=ti["src_poses"],
            src_Ks=ti["src_Ks"],
            cur_invK=ti["cur_invK"],
            min_depth=ti["min_depth"],
            max_depth=ti["max_depth"],
        )
        res = {}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            # A5: dot-product volume
            cv = ref["cost_volume"].CostVolumeManager(h, w, num_depth_bins=D)
            vol, low, planes, _ = cv(**common)
            res["dot_volume"] = vol.numpy()
            res["dot_lowest"] = low.numpy()
            res["planes"] = planes[:, :, 0, 0].numpy()
            # A2/A3/A4 intermediates for plane 3
            dp = planes[:, 3:4]
            uv_scale = torch.tensor([1 / w, 1 / h]).view(1, 1, 1, 2)
            wp, depths, warped, mask = cv.warp_features(
                ti["src_feats"], ti["src_extrinsics"], ti["src_Ks"], ti["cur_invK"], dp, b, k, 16, uv_scale
            )
            res["p3_world_points"] = wp.numpy()
            res["p3_depths"] = depths.numpy()
            res["p3_warped"] = warped.numpy()
            res["p3_mask"] = mask.numpy()
            # C2: metadata-MLP volume
            fv = ref["feature_volume"].FeatureVolumeManager(
                h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1], matching_dim_size=16, num_source_views=k
            )
            set_formula_weights(fv.mlp, 11 + seed, scale_mult=1.0)
            vol, low, _, m = fv(**common, return_mask=True)
            res["mlp_volume"] = vol.numpy()
            res["mlp_lowest"] = low.numpy()
            res["mlp_mask_slow"] = m.numpy()
            # C1: mesh-hint volume, slow and fast
            hv = ref["mesh_hint_volume"].FeatureMeshHintVolumeManager(
                h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1], matching_dim_size=16, num_source_views=k
            )
            set_formula_weights(hv.mlp, 11 + seed, scale_mult=1.0)
            set_formula_weights(hv.hint_mlp, 77 + seed, scale_mult=1.0)
            vol, low, _, m = hv(**common, cv_depth_hint_dict={n: v.clone() for n, v in hint.items()}, return_mask=True)
            res["hint_volume"] = vol.numpy()
            res["hint_lowest"] = low.numpy()
            res["hint_mask_slow"] = m.numpy()
            fast = hv.to_fast()
            volf, lowf, _, mf = fast(
                **common, cv_depth_hint_dict={n: v.clone() for n, v in hint.items()}, return_mask=True
            )
            res["hint_volume_fast"] = volf.numpy()
            res["hint_mask_fast"] = mf.numpy()
            # B1
            pd, rm, tm = ref["geometry_utils"].pose_distance(ti["src_poses"].view(-1, 4, 4))
            res["pose_dist"] = torch.stack([pd, rm, tm], 0).numpy()
        res["meta"] = np.array([b, k, h, w, D, seed, int(empty), int(behind)], dtype=np.int64)
        np.savez_compressed(os.path.join(out, f"volume_{name}.npz"), **res)
        print(f"volume_{name}: slow-vs-fast max diff {np.abs(res['hint_volume'] - res['hint_volume_fast']).max():.2e}")


def pixel_planes(b, D, h, w, min_depth, max_depth, seed):
    """Depth planes that vary over the image (the optional depth_planes_bdhw argument of the volume managers): the
    log-spaced list scaled by a smooth per-pixel factor in [0.8, 1.25] and a little noise; stored in the fixture."""
    from doubletake_amd.utils import synthetic as syn

    ramp = np.linspace(0, 1, D, dtype=np.float32).reshape(1, D, 1, 1)
    mn = np.asarray(min_depth, np.float32).reshape(-1, 1, 1, 1)
    mx = np.asarray(max_depth, np.float32).reshape(-1, 1, 1, 1)
    base = np.exp(np.log(mn) + np.log(mx / mn) * ramp).astype(np.float32)
    ys = np.linspace(-1, 1, h, dtype=np.float32).reshape(1, 1, h, 1)
    xs = np.linspace(-1, 1, w, dtype=np.float32).reshape(1, 1, 1, w)
    jitter = syn.hash_u01((b, D, h, w), seed).astype(np.float32)
    scale = np.exp(np.float32(0.2) * (np.sin(np.float32(2.1) * xs + ramp) * np.cos(np.float32(1.7) * ys))) * (
        np.float32(0.98) + np.float32(0.04) * jitter)
    return np.ascontiguousarray(np.broadcast_to(base, (b, D, h, w)) * scale).astype(np.float32)


def gen_volume_variants(ref, out):
    """Shapes the tuned kernels do not take, run on the reference: per-pixel depth_planes_bdhw (16 channels) and
    matching_dim_size = 8 / 24 -- dot, metadata-MLP and mesh-hint volumes (slow and fast managers)."""
    import torch
    from doubletake_amd.utils import synthetic as syn

    cases = {
        # name: (b, k, h, w, D, C, seed, per-pixel planes, behind_view)
        "pp_k3": (1, 3, 24, 32, 8, 16, 21, True, False),
        "pp_k2_b2": (2, 2, 19, 27, 5, 16, 22, True, True),
        "c8_k2": (1, 2, 24, 32, 8, 8, 23, False, False),
        "c24_k3_pp": (1, 3, 20, 28, 6, 24, 24, True, False),
    }
    res = {}
    for name, (b, k, h, w, D, C, seed, pp, behind) in cases.items():
        inp = syn.volume_inputs(b, k, h, w, C, seed, behind_view=behind)
        ti = {n: t(v) for n, v in inp.items()}
        hint = {n: ti[n] for n in ("depth_hint_b1hw", "depth_hint_mask_b1hw", "sampled_weights_b1hw")}
        common = dict(cur_feats=ti["cur_feats"], src_feats=ti["src_feats"], src_extrinsics=ti["src_extrinsics"],
                      src_poses=ti["src_poses"], src_Ks=ti["src_Ks"], cur_invK=ti["cur_invK"], min_depth=ti["min_depth"],
                      max_depth=ti["max_depth"])
        if pp:
            res[f"{name}/planes_bdhw"] = pixel_planes(b, D, h, w, inp["min_depth"], inp["max_depth"], 900 + seed)
            common["depth_planes_bdhw"] = t(res[f"{name}/planes_bdhw"])
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            cv = ref["cost_volume"].CostVolumeManager(h, w, num_depth_bins=D)
            vol, low, planes, _ = cv(**common)
            res[f"{name}/dot_volume"], res[f"{name}/dot_lowest"] = vol.numpy(), low.numpy()
            fv = ref["feature_volume"].FeatureVolumeManager(
                h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1], matching_dim_size=C, num_source_views=k)
            set_formula_weights(fv.mlp, 11 + seed, scale_mult=1.0)
            vol, low, _, m = fv(**common, return_mask=True)
            res[f"{name}/mlp_volume"], res[f"{name}/mlp_lowest"], res[f"{name}/mlp_mask"] = vol.numpy(), low.numpy(), m.numpy()
            hv = ref["mesh_hint_volume"].FeatureMeshHintVolumeManager(
                h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1], matching_dim_size=C, num_source_views=k)
            set_formula_weights(hv.mlp, 11 + seed, scale_mult=1.0)
            set_formula_weights(hv.hint_mlp, 77 + seed, scale_mult=1.0)
            vol, low, _, m = hv(**common, cv_depth_hint_dict={n: v.clone() for n, v in hint.items()}, return_mask=True)
            res[f"{name}/hint_volume"], res[f"{name}/hint_lowest"] = vol.numpy(), low.numpy()
            res[f"{name}/hint_mask_slow"] = m.numpy()
            if C == 16:  # (the reference's to_fast() rebuilds the manager with the default 16 channels)
                volf, lowf, _, mf = hv.to_fast()(
                    **common, cv_depth_hint_dict={n: v.clone() for n, v in hint.items()}, return_mask=True)
                res[f"{name}/hint_volume_fast"], res[f"{name}/hint_mask_fast"] = volf.numpy(), mf.numpy()
        res[f"{name}/meta"] = np.array([b, k, h, w, D, C, seed, int(pp), int(behind)], dtype=np.int64)
        print(f"volume_variants {name}: hint volume range {res[f'{name}/hint_volume'].min():.3f} .. "
              f"{res[f'{name}/hint_volume'].max():.3f}, lowest {res[f'{name}/hint_lowest'].min():.3f} .. "
              f"{res[f'{name}/hint_lowest'].max():.3f}")
    np.savez_compressed(os.path.join(out, "volume_variants.npz"), **res)


def gen_volume_fullsize(ref, out):
    """cfg1 / cfg2 full-size checksums."""
    import torch
    from doubletake_amd.utils import synthetic as syn

    cfgs = {"cfg1": (1, 2, 64, 80, 32, 101), "cfg2": (1, 7, 120, 160, 64, 102)}
    res = {}
    for name, (b, k, h, w, D, seed) in cfgs.items():
        inp = syn.volume_inputs(b, k, h, w, 16, seed)
        ti = {n: t(v) for n, v in inp.items()}
        hint = {n: ti[n] for n in ("depth_hint_b1hw", "depth_hint_mask_b1hw", "sampled_weights_b1hw")}
        common = {n: ti[n] for n in ("cur_feats", "src_feats", "src_extrinsics", "src_poses", "src_Ks", "cur_invK", "min_depth", "max_depth")}
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            cv = ref["cost_volume"].CostVolumeManager(h, w, num_depth_bins=D)
            vol, low, planes, _ = cv(**common)
            for kk, vv in checksum(vol.numpy()).items():
                res[f"{name}_dot_{kk}"] = vv
            hv = ref["mesh_hint_volume"].FeatureMeshHintVolumeManager(
                h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1], matching_dim_size=16, num_source_views=k
            )
            set_formula_weights(hv.mlp, 11 + seed)
            set_formula_weights(hv.hint_mlp, 77 + seed)
            vol, low, _, _ = hv(**common, cv_depth_hint_dict=hint)
            for kk, vv in checksum(vol.numpy()).items():
                res[f"{name}_hint_{kk}"] = vv
        res[f"{name}_meta"] = np.array([b, k, h, w, D, seed], dtype=np.int64)
        print(name, "done")
    np.savez_compressed(os.path.join(out, "volume_fullsize_checksums.npz"), **res)


#: whole-model cases at the BASELINE.json shapes (SURVEY.md section 8 cfg table): name -> (b, K, h, w, D, seed, decoder)
MODEL_FULLSIZE_CASES = {
    "cfg2_small": (1, 7, 120, 160, 64, 201, "skip"),      # configs[1]: DoubleTake-small 640x480
    "cfg2_full": (1, 7, 120, 160, 64, 202, "unet_pp"),    # the full model at the same frame size
    "cfg3_full_b8": (8, 7, 96, 128, 64, 203, "unet_pp"),  # configs[2]: full model, 512x384, batch 8
    "cfg3_small_b8": (8, 7, 96, 128, 64, 204, "skip"),
    "cfg4_small": (1, 7, 96, 128, 64, 205, "skip"),       # configs[3]: incremental mode is batch 1 at 512x384
    "cfg5_full_d96": (2, 7, 128, 96, 96, 206, "unet_pp"),  # configs[4]: portrait 384x512, 96 planes (CVEncoder in-ch = 96)
    "cfg5_small_d96": (2, 7, 128, 96, 96, 207, "skip"),
}
#: UNet++ nodes whose outputs are stored (reference networks.py:65-85: X_ij = in_conv_ij(cat(right, diag[, up])))
PP_PROBE_NODES = ("in_conv_31", "in_conv_22", "in_conv_13", "in_conv_01", "in_conv_02", "in_conv_03", "in_conv_04")
PP_LOG_STD = 0.4
MODEL_ENC_WIDTHS = {"skip": [64, 64, 128, 256, 512], "unet_pp": [24, 48, 64, 160, 256]}


def gen_model_fullsize(ref, out):
    """Whole hot path at full size through the reference's own modules: FeatureMeshHintVolumeManager (loop) ->
    CVEncoder -> SkipDecoderRegression | DepthDecoderPP -> exp, i.e. doubletake_model.py:379-418 without the image
    encoders.  Stores sum / abs-sum / min / max / 256 probes of every output (tests/test_model_fullsize_gpu.py)."""
    import torch
    from doubletake_amd.utils import synthetic as syn

    N, NF = ref["networks"], ref["networks_fast"]
    res = {}
    for name, (b, k, h, w, D, seed, dec_name) in MODEL_FULLSIZE_CASES.items():
        inp = syn.volume_inputs(b, k, h, w, 16, seed)
        ti = {n: t(v) for n, v in inp.items()}
        hint = {n: ti[n] for n in ("depth_hint_b1hw", "depth_hint_mask_b1hw", "sampled_weights_b1hw")}
        common = {n: ti[n] for n in ("cur_feats", "src_feats", "src_extrinsics", "src_poses", "src_Ks", "cur_invK", "min_depth", "max_depth")}
        enc = MODEL_ENC_WIDTHS[dec_name]
        pyr = [t(f) for f in syn.prior_pyramid(b, enc, 2 * h, 2 * w, seed + 50)]
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            hv = ref["mesh_hint_volume"].FeatureMeshHintVolumeManager(
                h, w, num_depth_bins=D, mlp_channels=[202, 128, 128, 1], matching_dim_size=16, num_source_views=k)
            set_formula_weights(hv.mlp, seed + 1)
            set_formula_weights(hv.hint_mlp, seed + 2)
            vol, low, _, mask = hv(**common, cv_depth_hint_dict={n: v.clone() for n, v in hint.items()}, return_mask=True)
            cve = N.CVEncoder(num_ch_cv=D, num_ch_enc=enc[1:], num_ch_outs=[64, 128, 256, 384])
            set_formula_weights(cve, seed + 3)
            cv_out = cve(vol, pyr[1:])
            pp_nodes = {}
            if dec_name == "skip":
                dec = NF.SkipDecoderRegression([enc[0]] + [64, 128, 256, 384])
                set_formula_weights(dec, seed + 4)
                dout = dec([pyr[0]] + cv_out)
            else:
                # Round 3 (VERDICT r2 "weak" #1): with scale_mult 0.7 the UNet++ outputs spanned log-depth < 0.1, so the
                # 1e-3 / 5e-4 tolerances were ~1 % of the signal.  Now: variance-preserving weights (scale_mult 1.0: O(1)
                # activations in every node), and the four 1x1 head convs are re-scaled / re-centred so that log depth has
                # zero mean and std PP_LOG_STD at every scale, i.e. depth spans roughly 0.3 .. 4 m.  The gains and biases
                # are data of the fixture (stored below); the test applies them to its own module.
                dec = N.DepthDecoderPP([enc[0]] + [64, 128, 256, 384])
                set_formula_weights(dec, seed + 4, scale_mult=1.0)
                hooks = [dec.convs[n].register_forward_hook(lambda m, i, o, n=n: pp_nodes.__setitem__(n, o)) for n in PP_PROBE_NODES]
                raw = dec([pyr[0]] + cv_out)
                for hk in hooks:
                    hk.remove()
                gains, biases = [], []
                for i in range(4):
                    head = dec.convs[f"output_{i}"][1]
                    ld = raw[f"log_depth_pred_s{i}_b1hw"]
                    gain = np.float32(np.clip(PP_LOG_STD / max(float(ld.std()), 1e-6), 0.25, 16.0))
                    head.weight.mul_(float(gain))
                    # new log depth = gain * (ld - old_bias) + new_bias; choose new_bias so that the mean is zero
                    new_bias = np.float32(-float(gain) * (float(ld.mean()) - float(head.bias[0])))
                    head.bias.fill_(float(new_bias))
                    gains.append(gain)
                    biases.append(new_bias)
                res[f"{name}|pp_head_gain"] = np.array(gains, dtype=np.float32)
                res[f"{name}|pp_head_bias"] = np.array(biases, dtype=np.float32)
                dout = dec([pyr[0]] + cv_out)
        outs = {"volume": vol, "lowest_cost": low, "mask_sum": mask.float().sum(1) if mask.dim() == 4 else mask.float()}
        for i, o in enumerate(cv_out):
            outs[f"cv_feat{i}"] = o
        for n, o in pp_nodes.items():   # UNet++ node outputs X_ij (= in_conv_ij), the decoder's counterpart of cv_feat
            outs[f"pp_{n}"] = o
        for kk, v in dout.items():
            if kk.startswith("log_depth"):
                outs[kk] = v
                outs[kk.replace("log_", "")] = torch.exp(v)
        for on, ov in outs.items():
            for kk, vv in checksum(ov.numpy(), nprobe=256).items():
                res[f"{name}|{on}|{kk}"] = vv
        res[f"{name}|meta"] = np.array([b, k, h, w, D, seed], dtype=np.int64)
        print(f"model_fullsize {name}: depth_s0 range {float(outs['depth_pred_s0_b1hw'].min()):.3f} .. "
              f"{float(outs['depth_pred_s0_b1hw'].max()):.3f}" + "".join(
                  f" | {n} |x| mean {float(o.abs().mean()):.2f}" for n, o in pp_nodes.items()), flush=True)
    np.savez_compressed(os.path.join(out, "model_fullsize_checksums.npz"), **res)


def gen_networks(ref, out):
    import torch
    from doubletake_amd.utils import synthetic as syn

    res = {}
    L = ref["layers"]
    N = ref["networks"]
    NF = ref["networks_fast"]
    with torch.no_grad():
        # E1: BasicBlock, three flavours
        for name, (cin, cout, stride) in {"bb_same": (16, 16, 1), "bb_chg": (24, 16, 1), "bb_s2": (16, 32, 2)}.items():
            blk = L.BasicBlock(cin, cout, stride=stride)
            set_formula_weights(blk, 500 + cin + cout)
            x = syn.hash_normalish((2, cin, 14, 18), 900 + cin)
            res[f"{name}_out"] = blk(t(x)).numpy()
        # base pyramid 16x20 at matching res (level 0 of CVEncoder)
        h0, w0 = 16, 24
        # E2 + G1: small model (resnet18d widths)
        enc_small = [64, 64, 128, 256, 512]
        D = 8
        cve = N.CVEncoder(num_ch_cv=D, num_ch_enc=enc_small[1:], num_ch_outs=[64, 128, 256, 384])
        set_formula_weights(cve, 1234)
        vol = syn.hash_normalish((1, D, h0, w0), 4321)
        feats = syn.prior_pyramid(1, enc_small, 2 * h0, 2 * w0, 555)  # level 0 at 2x matching res
        cv_out = cve(t(vol), [t(f) for f in feats[1:]])
        for i, o in enumerate(cv_out):
            res[f"cve_small_out{i}"] = o.numpy()
        dec = NF.SkipDecoderRegression([enc_small[0]] + [64, 128, 256, 384])
        set_formula_weights(dec, 2345)
        dout = dec([t(feats[0])] + cv_out)
        for kname, v in dout.items():
            res[f"skip_{kname}"] = v.numpy()
        # E2 + F1: full model (EfficientNetV2-S widths)
        enc_full = [24, 48, 64, 160, 256]
        cve2 = N.CVEncoder(num_ch_cv=D, num_ch_enc=enc_full[1:], num_ch_outs=[64, 128, 256, 384])
        set_formula_weights(cve2, 3456)
        feats2 = syn.prior_pyramid(1, enc_full, 2 * h0, 2 * w0, 666)
        cv_out2 = cve2(t(vol), [t(f) for f in feats2[1:]])
        for i, o in enumerate(cv_out2):
            res[f"cve_full_out{i}"] = o.numpy()
        dpp = N.DepthDecoderPP([enc_full[0]] + [64, 128, 256, 384])
        set_formula_weights(dpp, 4567, scale_mult=0.7)
        dout2 = dpp([t(feats2[0])] + cv_out2)
        for kname, v in dout2.items():
            res[f"pp_{kname}"] = v.numpy()
        # MLP alone
        mlp = N.MLP([10, 12, 12, 1], disable_final_activation=True)
        set_formula_weights(mlp, 99)
        xin = syn.hash_normalish((50, 10), 98)
        res["mlp_small_out"] = mlp(t(xin)).numpy()
    np.savez_compressed(os.path.join(out, "networks.npz"), **res)
    print("networks: keys", len(res))


class _Half:
    pass


def gen_tsdf(ref, out):
    import torch
    from doubletake_amd.utils import synthetic as syn

    T = ref["tsdf"]
    res = {}
    # T1: dims for 5 bounds dicts
    bounds_list = [
        dict(xmin=-4.0, xmax=4.0, ymin=-4.0, ymax=4.0, zmin=0.0, zmax=3.2),
        dict(xmin=-1.13, xmax=2.37, ymin=-0.41, ymax=1.95, zmin=-0.2, zmax=2.31),
        dict(xmin=0.0, xmax=2.56, ymin=0.0, ymax=2.24, zmin=0.0, zmax=2.24),
        dict(xmin=-10.0, xmax=-7.7, ymin=3.0, ymax=3.33, zmin=1.0, zmax=1.01),
        dict(xmin=-0.7, xmax=0.7, ymin=-0.7, ymax=0.7, zmin=-0.7, zmax=0.7),
    ]
    dims = []
    for bd in bounds_list:
        for vs in (0.04, 0.02):
            ts = T.TSDF.from_bounds(bd, vs)
            dims.append(list(ts.tsdf_values.shape))
    res["from_bounds_dims"] = np.array(dims, dtype=np.int64)
    res["from_bounds_list"] = np.array([[b[k] for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")] for b in bounds_list])
    # voxel coords of one small volume (fp16)
    ts = T.TSDF.from_bounds(bounds_list[1], 0.04)
    res["coords_case1_004"] = ts.voxel_coords_3hwd.numpy()

    # T3: integrate frames into a 64x56x56 @ 0.04 volume
    bd = dict(xmin=-1.28, xmax=1.28, ymin=-1.12, ymax=1.12, zmin=0.0, zmax=2.24)
    for tag, (vs, maxd, ext, H, W) in {
        "a": (0.04, 3.0, False, 120, 160),
        "b": (0.04, 3.0, True, 96, 128),
    }.items():
        ts = T.TSDF.from_bounds(bd, vs)
        fuser = T.TSDFFuser(ts, max_depth=maxd, use_gpu=False)
        depth, K, Tcw = syn.tsdf_frames(5, H, W, seed=3 if tag == "a" else 4, bounds=bd)
        depth = depth * np.float32(0.6)  # 0.9-1.8 m so surfaces land inside the small volume
        for nfr, upto in ((1, 1), (2, 2), (5, 5)):
            pass
        snap_at = {1, 2, 5}
        for f in range(5):
            fuser.integrate_depth(
                t(depth[f : f + 1]).half(),
                t(Tcw[f : f + 1]).half(),
                t(K[f : f + 1]).half(),
                extended_neg_truncation=ext,
            )
            if f + 1 in snap_at:
                res[f"int_{tag}_vals_{f + 1}"] = ts.tsdf_values.numpy().copy()
                res[f"int_{tag}_wts_{f + 1}"] = ts.tsdf_weights.numpy().copy()
                keys = np.array(sorted(ts.voxel_hashset.keys), dtype=np.int64).reshape(-1, 3)
                res[f"int_{tag}_active_{f + 1}"] = keys
        res[f"int_{tag}_dims"] = np.array(ts.tsdf_values.shape, dtype=np.int64)
        # T4: sample_tsdf at 256 points (weights and tsdf), fp32 on CPU
        pts = syn.hash_u01((256, 3), 4242).astype(np.float32)
        lo = np.array([bd["xmin"], bd["ymin"], bd["zmin"]], dtype=np.float32) - 0.1
        hi = np.array([bd["xmax"], bd["ymax"], bd["zmax"]], dtype=np.float32) + 0.1
        pts = lo + pts * (hi - lo)
        res[f"sample_{tag}_pts"] = pts
        res[f"sample_{tag}_weights"] = ts.sample_tsdf(t(pts), "weights").numpy()
        res[f"sample_{tag}_tsdf"] = ts.sample_tsdf(t(pts), "tsdf").numpy()
        print(f"tsdf {tag}: active {len(ts.voxel_hashset.keys)}, updated {(ts.tsdf_weights > 0).sum().item()}")
    np.savez_compressed(os.path.join(out, "tsdf.npz"), **res)


def gen_formats(ref, out):
    """Files written by the reference's own writers: a saved TSDF volume, two depth-cache pickles and a
    score sheet.  They are data fixtures for tests/test_formats.py."""
    import shutil

    import torch
    from doubletake_amd.utils import synthetic as syn

    T = ref["tsdf"]
    from doubletake.utils import generic_utils, metrics_utils

    fdir = os.path.join(out, "formats")
    shutil.rmtree(fdir, ignore_errors=True)
    os.makedirs(fdir)
    bd = dict(xmin=-0.64, xmax=0.64, ymin=-0.56, ymax=0.56, zmin=0.0, zmax=1.12)
    ts = T.TSDF.from_bounds(bd, 0.04)
    fuser = T.TSDFFuser(ts, max_depth=3.0, use_gpu=False)
    depth, K, Tcw = syn.tsdf_frames(2, 48, 64, seed=11, bounds=bd)
    depth = depth * np.float32(0.35)
    for f in range(2):
        fuser.integrate_depth(t(depth[f:f + 1]).half(), t(Tcw[f:f + 1]).half(), t(K[f:f + 1]).half())
    ts.save_tsdf(os.path.join(fdir, "ref_saved_tsdf.npz"))
    np.savez_compressed(os.path.join(fdir, "ref_saved_tsdf_inputs.npz"), depth=depth, K=K, Tcw=Tcw,
                        bounds=np.array([bd[k] for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")]))

    # depth cache: batch of two keyframes
    b, h, w = 2, 12, 16
    outputs = {
        "depth_pred_s0_b1hw": t(syn.hash_u01((b, 1, h, w), 21).astype(np.float32) + 0.5),
        "overall_mask_bhw": t(syn.hash_u01((b, h, w), 22).astype(np.float32)) > 0.3,
        "cv_confidence_b1hw": t(syn.hash_u01((b, 1, h, w), 23).astype(np.float32)),
    }
    eye = np.tile(np.eye(4, dtype=np.float32), (b, 1, 1))
    cur_data = {
        "frame_id_string": ["000012", "000031"],
        "K_full_depth_b44": t(eye * 2.0), "K_s0_b44": t(eye * 0.5), "cam_T_world_b44": t(eye),
    }
    src_data = {"frame_id_string": [["000010", "000029"], ["000008", "000027"]]}
    os.makedirs(os.path.join(fdir, "depth_cache"))
    generic_utils.cache_model_outputs(os.path.join(fdir, "depth_cache"), outputs, cur_data, src_data, 0, b)
    torch.save({k: v for k, v in outputs.items()}, os.path.join(fdir, "depth_cache_inputs.pt"))

    # score sheet
    avg = metrics_utils.ResultsAverager("golden_exp", "frame metrics")
    for i in range(3):
        gt = t(syn.hash_u01((1, 200), 31 + i).astype(np.float32) + 1.0)
        pr = gt * t(1.0 + 0.2 * (syn.hash_u01((1, 200), 41 + i).astype(np.float32) - 0.5))
        m = metrics_utils.compute_depth_metrics(gt, pr)
        avg.update_results({k: float(v) for k, v in m.items()})
    avg.compute_final_average()
    avg.output_json(os.path.join(fdir, "ref_scores.json"))
    np.savez(os.path.join(fdir, "ref_scores_inputs.npz"),
             names=np.array(list(avg.final_metrics.keys())), values=np.array([float(v) for v in avg.final_metrics.values()]))
    print("formats:", sorted(os.listdir(fdir)))


def main():
    which = set(sys.argv[1:]) or {"volume", "variants", "fullsize", "networks", "model_fullsize", "tsdf", "formats"}
    ref = import_reference()
    if "volume" in which:
        gen_volume(ref, OUT)
    if "variants" in which:
        gen_volume_variants(ref, OUT)
    if "fullsize" in which:
        gen_volume_fullsize(ref, OUT)
    if "networks" in which:
        gen_networks(ref, OUT)
    if "model_fullsize" in which:
        gen_model_fullsize(ref, OUT)
    if "tsdf" in which:
        gen_tsdf(ref, OUT)
    if "formats" in which:
        gen_formats(ref, OUT)


if __name__ == "__main__":
    main()
