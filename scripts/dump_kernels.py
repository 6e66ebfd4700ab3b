#!/usr/bin/env python3
"""Every dispatch of a rocprofv3 kernel trace (rocpd SQLite) in launch order: start, duration, grid, name.
    python scripts/dump_kernels.py gpurun_out/xxx/yyy_results.db [name-filter]"""
import re
import sqlite3
import sys

rows = list(sqlite3.connect(sys.argv[1]).cursor().execute(
    "select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start"))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
t0 = rows[0][1]
for name, s, e, gx, wx, vg, ag, lds in rows:
    k = re.sub(r"\(.*", "", name).replace("void ", "")
    if flt and flt not in k:
        continue
    print(f"{(s - t0) / 1e3:12.1f} {(e - s) / 1e3:8.2f} {gx // max(wx, 1):6d}x{wx:<5d} {vg + ag:4d} {lds:7d}  {k[:80]}")
