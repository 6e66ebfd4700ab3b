#!/usr/bin/env python3
"""Timeline of ONE bench step from a rocprofv3 kernel trace (rocpd SQLite): every dispatch between two
consecutive launches of the volume kernel, with grid, duration and the gap to the previous kernel's end.

    python scripts/step_timeline.py gpurun_out/prof/xxx_results.db [--step -2] > profiles/NAME_step_timeline.txt
"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, scratch_size "
                            "from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "cv_mlp_mfma_kernel" in r[0]]
    a, b = marks[which], marks[which + 1]
    step = rows[a:b]
    t0 = step[0][1]
    prev_end = None
    busy = 0
    print(f"# step of {len(step)} dispatches, {(rows[b][1] - t0) / 1e3:.1f} us start-to-start")
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} {'wgs':>6} {'wg':>5} {'vgpr':>5} {'lds':>7}  name")
    for name, s, e, gx, wx, vg, ag, lds, scr in step:
        k = re.sub(r"\(.*", "", name).replace("void ", "")[:70]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f} {gx // max(wx, 1):6d} {wx:5d} {vg + ag:5d} {lds:7d}  {k}")
        prev_end = e
        busy += e - s
    print(f"# kernel time {busy / 1e3:.1f} us, gaps {(rows[b][1] - t0 - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
