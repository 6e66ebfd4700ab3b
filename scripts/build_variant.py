#!/usr/bin/env python3
"""Build an experimental variant of the library next to the shipped one (A/B runs on the GPU box):

    python scripts/build_variant.py NAME "-DDT_WINO_PF=1 -DDT_WINO1_WPE=2"   -> doubletake_amd/_lib/variants/NAME.so
    DOUBLETAKE_HIP_LIB=doubletake_amd/_lib/variants/NAME.so python bench.py ...

Same flags as doubletake_amd/_build.py plus the given ones; objects go to a private directory."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from doubletake_amd import _build as B


def main():
    name, extra = sys.argv[1], (sys.argv[2].split() if len(sys.argv) > 2 else [])
    out_dir = os.path.join(B.LIBDIR, "variants")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-I", B.INCLUDE, "-c", os.path.join(B.CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, B._sources()))
    lib = os.path.join(out_dir, name + ".so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print(lib)


if __name__ == "__main__":
    main()
