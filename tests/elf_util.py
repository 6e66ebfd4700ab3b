"""Kernel resource metadata of the built library (test infrastructure).

hipcc embeds one clang offload bundle per translation unit in the ``.hip_fatbin`` section of libdoubletake_hip.so
(magic ``__CLANG_OFFLOAD_BUNDLE__``, then a table of (offset, size, target triple)).  ``kernel_notes(path)`` extracts the
gfx950 code objects and reads their AMDGPU metadata notes with llvm-readelf: name -> dict of the integer fields
(.vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size, .group_segment_fixed_size, ...).
"""
from __future__ import annotations

import os
import re
import shutil
import struct
import subprocess
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF_CANDIDATES = ("/opt/rocm/lib/llvm/bin/llvm-readelf", "llvm-readelf")


def readelf():
    for c in READELF_CANDIDATES:
        p = c if os.path.isabs(c) and os.path.exists(c) else shutil.which(c)
        if p:
            return p
    return None


def code_objects(path, arch="gfx950"):
    """The embedded device ELFs for ``arch`` as a list of bytes objects."""
    blob = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", blob, i + len(MAGIC))[0]
        q = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode("ascii", "replace")
            q += 24 + tlen
            if arch in triple and size:
                out.append(blob[i + off:i + off + size])
        pos = i + len(MAGIC)
    return out


def kernel_notes(path, arch="gfx950"):
    exe = readelf()
    if exe is None:
        raise RuntimeError("llvm-readelf not found")
    notes = {}
    with tempfile.TemporaryDirectory() as d:
        for j, co in enumerate(code_objects(path, arch)):
            f = os.path.join(d, f"co{j}.elf")
            open(f, "wb").write(co)
            txt = subprocess.run([exe, "--notes", f], capture_output=True, text=True, check=True).stdout
            # the metadata is YAML: "- .agpr_count: 0 \n .args: ... .name: kernel ... .vgpr_spill_count: 0"
            for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
                block = ".agpr_count:" + block
                m = re.search(r"^\s*\.name:\s*(\S+)", block, re.M)
                if not m:
                    continue
                fields = {k: int(v) for k, v in re.findall(r"^\s*\.(\w+):\s*(\d+)\s*$", block, re.M)}
                notes[m.group(1)] = fields
    return notes
