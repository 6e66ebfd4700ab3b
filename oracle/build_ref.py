#!/usr/bin/env python3
"""ORACLE build recipe (test infrastructure).  Compiles the reference's CPU marching cubes from the
sources where they lie under /root/reference into oracle/_ref/ (git-ignored, travels with gpurun).

    python oracle/build_ref.py

Needs /root/reference; on the GPU box the prebuilt oracle/_ref/dt_ref_mc.so is used as is.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MC = "/root/reference/src/doubletake/tools/marching_cubes"
OUT = os.path.join(HERE, "_ref")


def so_path():
    return os.path.join(OUT, "dt_ref_mc.so")


def build(verbose=False):
    if os.path.isfile(so_path()):
        return so_path()
    if not os.path.isdir(REF_MC):
        raise RuntimeError("reference sources not available; oracle/_ref cannot be built here")
    from torch.utils.cpp_extension import load

    os.makedirs(OUT, exist_ok=True)
    load(name="dt_ref_mc", sources=[os.path.join(HERE, "ref_mc_binding.cpp"), os.path.join(REF_MC, "marching_cubes_cpu.cpp")],
         extra_include_paths=[REF_MC], build_directory=OUT, verbose=verbose, is_python_module=True)
    return so_path()


def load_module():
    """Import the prebuilt module (building it first when the reference is present)."""
    import importlib.util

    import torch  # noqa: F401

    path = build()
    spec = importlib.util.spec_from_file_location("dt_ref_mc", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
