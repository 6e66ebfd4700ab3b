#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into the per-kernel table that
`rocprofv3 --stats` prints: calls, total / average / min / max duration, share of GPU time.

    python scripts/summarize_rocpd.py gpurun_out/prof/bench_results.db profiles/NAME_kernel_stats.csv [--skip-first N]
"""
import collections
import csv
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start"))[skip:]
    agg = collections.OrderedDict()
    for name, s, e, gx, wx, vg, ag, lds in rows:
        k = re.sub(r"\(.*", "", name).replace("void ", "")
        a = agg.setdefault(k, dict(calls=0, total=0, mn=1 << 62, mx=0, vgpr=vg, agpr=ag, lds=lds))
        d = e - s
        a["calls"] += 1
        a["total"] += d
        a["mn"] = min(a["mn"], d)
        a["mx"] = max(a["mx"], d)
    tot = sum(a["total"] for a in agg.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "LDS"])
        for k, a in sorted(agg.items(), key=lambda x: -x[1]["total"]):
            w.writerow([k, a["calls"], a["total"], round(a["total"] / a["calls"], 1), round(100.0 * a["total"] / tot, 3),
                        a["mn"], a["mx"], a["vgpr"], a["agpr"], a["lds"]])
    print(f"{len(rows)} dispatches, {tot / 1e6:.3f} ms of kernel time -> {out}")


if __name__ == "__main__":
    main()
