"""Build the C-ABI shared library (hipcc, gfx950 only) in-tree.

    python -m doubletake_amd._build [--force]

Output: doubletake_amd/_lib/libdoubletake_hip.so (+ build_hash.txt).  The .so is git-ignored
but travels with the working tree (gpurun snapshot), so the GPU box does not rebuild unless
the sources changed.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libdoubletake_hip.so")
HASHFILE = os.path.join(LIBDIR, "build_hash.txt")
# -fno-slp-vectorize: hipcc otherwise packs adjacent scalar fp32 ops into v_pk_mul/v_pk_fma/v_pk_add_f32, which
# cost extra issue cycles beside MFMAs on gfx950 (MI355X_MICROARCH.md, "price of one filler beside MFMAs");
# measured on the volume kernel: 0.807 -> 0.795 ms
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-slp-vectorize"]
# per-file extras.  tsdf.hip restates the reference's half pipeline op by op: fusing a multiply
# and an add of two different reference ops into one FMA would drop a rounding, so contraction is
# off for the whole file (hipcc's default is fast-honor-pragmas).
EXTRA_FLAGS = {"tsdf.hip": ["-ffp-contract=off"]}
# experiment hook: extra -D flags for every file (e.g. DT_EXTRA_CFLAGS="-DDT_MLP_PRIO=1")
FLAGS += os.environ.get("DT_EXTRA_CFLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _hash():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(INCLUDE, "doubletake_hip.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def is_current():
    return os.path.isfile(LIB) and os.path.isfile(HASHFILE) and open(HASHFILE).read().strip() == _hash()


def build(force=False, verbose=True):
    if not force and is_current():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(HASHFILE, "w") as f:
        f.write(_hash() + "\n")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
