#!/usr/bin/env python3
"""How the kernels of two HIP streams share the GPU: from a rocprofv3 kernel trace (csv) of `bench.py --streams 2`, per
kernel name the mean duration when it ran alone vs while a dominant (volume) kernel of the other stream was resident, and the
fraction of wall time covered by 0 / 1 / 2+ kernels.

    python scripts/overlap_report.py gpurun_out/xxx/b_kernel_trace.csv
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("dt::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "0"))))
    rows.sort()
    vols = [(s, e, q) for s, e, n, q in rows if n.startswith("cv_mlp_mfma_kernel")]
    # steady-state window: from the 10th to the last-but-5th volume kernel
    t_lo, t_hi = vols[10][0], vols[-5][0]
    rows = [r for r in rows if t_lo <= r[0] < t_hi]
    vols = [v for v in vols if t_lo <= v[0] < t_hi]
    print(f"window {1e-3 * (t_hi - t_lo):.0f} us, {len(vols)} volume kernels -> {1e-3 * (t_hi - t_lo) / len(vols):.1f} us per frame; "
          f"volume kernel mean {sum(e - s for s, e, _ in vols) / len(vols) / 1e3:.1f} us")
    # coverage by concurrency level
    ev = sorted([(s, 1) for s, e, n, q in rows] + [(e, -1) for s, e, n, q in rows])
    cover = defaultdict(int)
    depth, last = 0, t_lo
    for t, d in ev:
        cover[min(depth, 3)] += max(0, min(t, t_hi) - max(last, t_lo))
        depth += d
        last = t
    tot = sum(cover.values())
    print("wall-time share by number of resident kernels: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {100.0 * v / tot:.1f} %" for k, v in sorted(cover.items())))
    # volume-vs-volume overlap
    vv = 0
    for i, (s, e, q) in enumerate(vols):
        for s2, e2, q2 in vols[i + 1:i + 3]:
            vv += max(0, min(e, e2) - max(s, s2))
    print(f"volume kernels overlapping each other: {100.0 * vv / tot:.1f} % of wall time")
    # per kernel: alone vs under a foreign volume kernel
    stat = defaultdict(lambda: [0, 0.0, 0, 0.0])
    for s, e, n, q in rows:
        if n.startswith("cv_mlp_mfma_kernel"):
            continue
        under = any(vs < e and ve > s and vq != q for vs, ve, vq in vols)
        st = stat[n]
        if under:
            st[2] += 1
            st[3] += e - s
        else:
            st[0] += 1
            st[1] += e - s
    print(f"{'kernel':60s} {'n alone':>8} {'us':>8} {'n under vol':>12} {'us':>8}")
    for n, (a, ta, b, tb) in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
        print(f"{n[:60]:60s} {a:8d} {ta / max(a, 1) / 1e3:8.1f} {b:12d} {tb / max(b, 1) / 1e3:8.1f}")


if __name__ == "__main__":
    main()
