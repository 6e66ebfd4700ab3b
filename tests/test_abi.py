"""The C-ABI library builds, loads without a GPU and exports every symbol include/doubletake_hip.h
declares; argument validation fails loudly before any launch.  CPU only (no compute calls)."""
import ctypes
import os
import re

import pytest

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "doubletake_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from doubletake_amd import _abi

    L = _abi.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in doubletake_hip.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_abi.SIGNATURES) == syms
    # the ABI version the bindings were written against == the header's == what the built library reports
    header_version = int(re.search(r"#define\s+DT_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert L.dt_version() == header_version == _abi.ABI_VERSION


def test_argument_errors_are_reported_not_launched():
    from doubletake_amd import _abi

    L = _abi.lib()
    assert L.dt_cv_setup_f32(None, None, None, None, None, None, 1, 7, 64, None, None) != 0
    assert b"null pointer" in L.dt_last_error()
    assert L.dt_cv_dot_f32(None, None, None, None, 1, 7, 12, 4, 4, 8, None) != 0
    assert b"channels" in L.dt_last_error() or b"null" in L.dt_last_error()
    n = ctypes.c_int()
    assert L.dt_cv_mlp_pack_floats(16, ctypes.byref(n), None, None, None) != 0   # the fused kernel takes 1..15 source views
    from doubletake_amd.modules import mlp_pack

    assert L.dt_cv_mlp_pack_floats(9, ctypes.byref(n), None, None, None) == 0 and n.value == mlp_pack.dyn_steps_total(9) * 256 == 104 * 256
    assert L.dt_cv_mlp_pack_floats(7, ctypes.byref(n), None, None, None) == 0 and n.value == 81 * 256
    # the paired-metadata step layout is defined twice -- dt::mlp_view_step_base in the kernel, mlp_pack.view_step_base on the
    # host -- and must agree for every view count the fused kernel takes
    for K in range(1, 16):
        assert L.dt_cv_mlp_pack_floats(K, ctypes.byref(n), None, None, None) == 0
        assert n.value == mlp_pack.dyn_steps_total(K) * 256, K
    assert L.dt_conv_pack_floats(64, 64, 3) == 64 * 64 * 9
    d = _abi.ConvDesc()
    assert L.dt_conv2d_f32(ctypes.byref(d), None, None, None, None, None, None, None, None) != 0
    with pytest.raises(_abi.DoubletakeHipError):
        _abi.check(1, "x")


def test_no_cpu_fallback_in_product_modules():
    import torch

    from doubletake_amd import _abi
    from doubletake_amd.modules.cost_volume import CostVolumeManager
    from doubletake_amd.modules.layers import BasicBlock

    m = CostVolumeManager(8, 8, 4)
    z = torch.zeros
    with pytest.raises(_abi.DoubletakeHipError):
        m(z(1, 16, 8, 8), z(1, 2, 16, 8, 8), z(1, 2, 4, 4), z(1, 2, 4, 4), z(1, 2, 4, 4), z(1, 4, 4), z(1), z(1))
    with pytest.raises(_abi.DoubletakeHipError):
        BasicBlock(32, 32)(z(1, 32, 8, 8))


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "doubletake_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_model_accepts_reference_opts_object():
    """DepthModelCVHint(opts) as the reference constructs it (experiment_modules/doubletake_model.py:84-204)."""
    import types

    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint

    opts = types.SimpleNamespace(image_height=192, image_width=256, image_encoder_name="resnet18d", depth_decoder_name="skip",
                                 matching_num_depth_bins=32, matching_scale=1, matching_feature_dims=16, model_num_views=5,
                                 min_matching_depth=0.25, max_matching_depth=5.0, matching_encoder_type="resnet",
                                 cv_encoder_type="multi_scale_encoder", feature_volume_type="mlp_mesh_hint_feature_volume",
                                 loss_type="log_l1")
    m = DepthModelCVHint(opts)
    assert m.cost_volume.num_source_views == 4 and m.cost_volume.num_depth_bins == 32
    assert (m.cost_volume.matching_height, m.cost_volume.matching_width) == (48, 64)
    keys = list(m.state_dict().keys())
    assert any(k.startswith("matching_model.net.0.") for k in keys) and any(k.startswith("cost_volume.mlp.") for k in keys)
    assert any(k.startswith("cost_volume_net.") for k in keys) and any(k.startswith("depth_decoder.") for k in keys)
    opts.feature_volume_type = "mlp_feature_volume"
    with pytest.raises(ValueError):  # doubletake_model.py:174-177: DoubleTake only builds the mesh-hint volume
        DepthModelCVHint(opts)


def test_model_selection_follows_reference_model_utils(tmp_path):
    """utils/model_utils.py:10-35 + sr_depth_model.py:186-194: model_type -> class, feature_volume_type -> manager,
    fast_cost_volume -> to_fast(), checkpoint loading by state_dict with the reference's key names."""
    import types

    import torch

    from doubletake_amd.experiment_modules.doubletake_model import DepthModel, DepthModelCVHint
    from doubletake_amd.modules import cost_volume as cv
    from doubletake_amd.utils import model_utils as mu

    base = dict(image_height=64, image_width=96, model_num_views=3, matching_num_depth_bins=16, fast_cost_volume=False,
                load_weights_from_checkpoint=None)
    o = types.SimpleNamespace(model_type="depth_model", feature_volume_type="simple_cost_volume", **base)
    assert mu.get_model_class(o) is DepthModel
    m = mu.load_model_inference(o, DepthModel)
    assert type(m.cost_volume) is cv.CostVolumeManager and not any(k.startswith("cost_volume.mlp") for k in m.state_dict())
    o.feature_volume_type = "mlp_feature_volume"
    m = mu.load_model_inference(o, DepthModel)
    assert type(m.cost_volume) is cv.FeatureVolumeManager and not hasattr(m.cost_volume, "hint_mlp")
    o.fast_cost_volume = True
    assert type(mu.load_model_inference(o, DepthModel).cost_volume) is cv.FastFeatureVolumeManager
    o.feature_volume_type = "mlp_mesh_hint_feature_volume"
    with pytest.raises(ValueError):
        DepthModel(o)
    o.model_type = "cv_hint_depth_model"
    assert mu.get_model_class(o) is DepthModelCVHint
    assert type(mu.load_model_inference(o, DepthModelCVHint).cost_volume) is cv.FastFeatureMeshHintVolumeManager
    o.model_type = "other"
    with pytest.raises(ValueError):
        mu.get_model_class(o)
    # checkpoint round trip: a reference checkpoint also carries keys of modules outside the hot path (encoder.*)
    o.model_type, o.fast_cost_volume = "cv_hint_depth_model", False
    src = DepthModelCVHint(o)
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.fill_(0.001 * (i + 1))
    sd = dict(src.state_dict())
    sd["encoder.conv1.weight"] = torch.zeros(3)
    path = tmp_path / "ckpt.pt"
    torch.save({"state_dict": sd}, path)
    o.load_weights_from_checkpoint = str(path)
    got = mu.load_model_inference(o, DepthModelCVHint)
    assert got.unused_checkpoint_keys == ["encoder.conv1.weight"]
    for (n1, a), (n2, b) in zip(src.state_dict().items(), got.state_dict().items()):
        assert n1 == n2 and torch.equal(a, b)
    del sd["cost_volume.mlp.net.0.weight"]
    torch.save({"state_dict": sd}, path)
    with pytest.raises(RuntimeError):
        mu.load_model_inference(o, DepthModelCVHint)


def test_lightning_style_checkpoint_with_foreign_hyper_parameters_loads(tmp_path):
    """reference checkpoints are Lightning files whose ``hyper_parameters`` hold a pickled
    ``doubletake.options`` object (doubletake_model.py:116).  The loader must read ``state_dict`` from such a file
    without the defining module being importable, and must not execute foreign reduce callables."""
    import sys
    import types

    import torch

    from doubletake_amd.experiment_modules.doubletake_model import DepthModelCVHint
    from doubletake_amd.utils import model_utils as mu

    o = types.SimpleNamespace(model_type="cv_hint_depth_model", feature_volume_type="mlp_mesh_hint_feature_volume",
                              image_height=64, image_width=96, model_num_views=3, matching_num_depth_bins=16,
                              fast_cost_volume=False, load_weights_from_checkpoint=None)
    src = DepthModelCVHint(o)
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.fill_(0.002 * (i + 1))
    # a throw-away module standing in for doubletake.options; gone again before the file is read
    mod = types.ModuleType("fake_doubletake_options")
    exec("class Options:\n    def __init__(self):\n        self.image_height = 64\n        self.name = 'x'\n"
         "def evil(*a):\n    raise SystemExit('reduce callable executed')\n"
         "class Trap:\n    def __reduce__(self):\n        return (evil, (1,))\n", mod.__dict__)
    sys.modules["fake_doubletake_options"] = mod
    try:
        path = tmp_path / "lightning.ckpt"
        torch.save({"epoch": 3, "global_step": 10, "pytorch-lightning_version": "1.8.4",
                    "state_dict": dict(src.state_dict()), "hyper_parameters": {"opts": mod.Options()},
                    "callbacks": {"trap": mod.Trap()}, "optimizer_states": [{"state": {}, "param_groups": []}]}, path)
    finally:
        del sys.modules["fake_doubletake_options"]
    with pytest.raises(Exception):  # the stock loader refuses the file (torch >= 2.6 defaults to weights_only=True)
        torch.load(path, map_location="cpu")
    state = mu.read_checkpoint_state_dict(path)
    assert set(state) == set(src.state_dict())
    o.load_weights_from_checkpoint = str(path)
    got = mu.load_model_inference(o, DepthModelCVHint)
    for (n1, a), (n2, b) in zip(src.state_dict().items(), got.state_dict().items()):
        assert n1 == n2 and torch.equal(a, b)
    # a plain state-dict file still loads; a file without tensors is rejected
    torch.save(dict(src.state_dict()), tmp_path / "plain.pt")
    assert set(mu.read_checkpoint_state_dict(tmp_path / "plain.pt")) == set(src.state_dict())
    torch.save({"state_dict": {"a": 1}}, tmp_path / "bad.pt")
    with pytest.raises(RuntimeError):
        mu.read_checkpoint_state_dict(tmp_path / "bad.pt")


def test_legacy_format_hostile_pickle_never_runs_its_reduce_callable(tmp_path):
    """torch's legacy (non-zip) reader calls ``pickle_module.load`` directly -- for the magic number,
    protocol version, sys_info, storage keys -- so every entry point of the pickle namespace handed to ``torch.load``
    must be the restricted unpickler.  A file that is just ``pickle.dumps(obj)`` with a hostile ``__reduce__`` must be
    rejected without the callable having run; a legacy-format checkpoint and a TorchScript-looking zip behave as stated."""
    import pickle
    import zipfile

    import torch

    from doubletake_amd.utils import model_utils as mu

    marker = tmp_path / "executed.txt"

    class Hostile:
        def __reduce__(self):
            import pathlib

            return (pathlib.Path.write_text, (pathlib.Path(str(marker)), "ran"))

    bad = tmp_path / "hostile.ckpt"
    bad.write_bytes(pickle.dumps(Hostile()))
    with pytest.raises(Exception):
        mu.read_checkpoint_state_dict(bad)
    assert not marker.exists(), "a reduce callable of the file was executed"
    # the same object hidden behind a valid legacy header (magic number, protocol, sys_info, then the payload)
    bad2 = tmp_path / "hostile_legacy.ckpt"
    with open(bad2, "wb") as f:
        pickle.dump(0x1950A86A20F9469CFC6C, f, protocol=2)
        pickle.dump(1001, f, protocol=2)
        pickle.dump(Hostile(), f, protocol=2)
    with pytest.raises(Exception):
        mu.read_checkpoint_state_dict(bad2)
    assert not marker.exists(), "a reduce callable of the file was executed"
    # an honest legacy-format file (torch.save(..., _use_new_zipfile_serialization=False)) still loads
    sd = {"a.weight": torch.arange(6.0).view(2, 3), "a.bias": torch.ones(2)}
    legacy = tmp_path / "legacy.pt"
    torch.save({"state_dict": sd}, legacy, _use_new_zipfile_serialization=False)
    got = mu.read_checkpoint_state_dict(legacy)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    # a zip that torch.load would hand to torch.jit.load is refused before torch.load sees it
    ts = tmp_path / "script.pt"
    with zipfile.ZipFile(ts, "w") as z:
        z.writestr("archive/constants.pkl", b"\x80\x02)." )
        z.writestr("archive/data.pkl", b"\x80\x02}.")
    with pytest.raises(RuntimeError, match="TorchScript"):
        mu.read_checkpoint_state_dict(ts)


def test_hot_path_kernels_do_not_spill():
    """the claim "no spills on the path" is checked against the ELF, not against a document.  The AMDGPU
    metadata notes of every gfx950 code object in the built library: the dominant kernel (every instantiation of the fused
    volume kernel the DoubleTake / SimpleRecon paths launch), the Winograd / direct / paired conv kernels, the head kernels and
    the TSDF integrate kernel must have no spilled VGPRs and no private (scratch) segment.  (SGPR spills go to lanes of a
    reserved VGPR, not to memory; they are listed, not forbidden.)"""
    import elf_util

    from doubletake_amd import _build

    if elf_util.readelf() is None:
        pytest.skip("llvm-readelf not available")
    if not os.path.isfile(_build.LIB):
        pytest.skip("library not built")
    notes = elf_util.kernel_notes(_build.LIB)
    assert len(notes) >= 60, len(notes)
    on_path = ("cv_mlp_mfma_kernelILb1ELi8ELb0", "cv_mlp_mfma_kernelILb0ELi8ELb0", "cv_mlp_mfma_kernelILb1ELi4ELb0",
               "cv_mlp_mfma_kernelILb0ELi4ELb0", "cv_mlp_mfma_kernelILb1ELi4ELb1", "cv_mlp_mfma_kernelILb0ELi4ELb1",
               "conv_wino_kernel", "conv_wino_heads_kernel", "conv_pair_kernel", "conv_mfma_kernel", "conv_mfma_wshare_kernel",
               "conv1x1_mfma_kernel", "head_mlp_kernel", "head_mlp_multi_kernel", "head_mlp_split_kernel", "tsdf_integrate_kernel",
               "cv_lowest_cost_kernel", "cv_mask_kernel", "mlp_plan_cost_kernel", "mlp_plan_bounds_kernel", "stem_conv_kernel",
               "maxblur", "instnorm")
    seen = {tag: 0 for tag in on_path}
    bad = []
    for name, f in notes.items():
        for tag in on_path:
            if tag in name:
                seen[tag] += 1
                if f.get("vgpr_spill_count", 0) != 0 or f.get("private_segment_fixed_size", 0) != 0:
                    bad.append((name, f.get("vgpr_count"), f.get("vgpr_spill_count"), f.get("private_segment_fixed_size")))
    assert all(n > 0 for n in seen.values()), {k: v for k, v in seen.items() if v == 0}
    assert not bad, bad
    assert not any("conv_wino_kernelILi4E" in n for n in notes)  # (the experimental 1024-thread instantiation is gone)
