#!/usr/bin/env python3
"""Refresh profiles/roofline_traffic.json -- the HBM bytes per launch of cv_mlp_mfma_kernel that bench.py reports as
roofline.traffic -- from a PMC summary written by scripts/pmc_summary.py.

    python scripts/make_roofline_traffic.py r3c        # reads profiles/r3c_pmc_summary.json
    python scripts/make_roofline_traffic.py r6z profiles/r6z_bench_kernel_stats.csv   # + the kernel's rocprofv3 average

The record carries the hash of the kernel source the counters were collected on; bench.py reports the figure only while
that hash equals the source it runs (so a stale pass can never be attributed to a newer kernel).  Run this after every
PMC pass over bench.py (the line kept quoting an older pass although a newer one existed)."""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    summ = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_summary.json")))
    name = next(k for k in summ if "cv_mlp_mfma_kernel" in k)
    rec = summ[name]
    fetch_kb, write_kb = rec["FETCH_SIZE"], rec["WRITE_SIZE"]
    src = open(os.path.join(REPO, "doubletake_amd", "csrc", "cv_mlp_mfma.hip"), "rb").read()
    rocprof_ns = rocprof_calls = stats_file = None
    if len(sys.argv) > 2:  # kernel stats of a rocprofv3 --kernel-trace --stats run of bench.py on the same library
        import csv

        stats_file = sys.argv[2]
        for row in csv.DictReader(open(os.path.join(REPO, stats_file) if not os.path.isabs(stats_file) else stats_file)):
            if "cv_mlp_mfma_kernel" in row["Name"]:
                rocprof_ns, rocprof_calls = float(row["AverageNs"]), int(row["Calls"])
                break
    out = {
        "profile_tag": tag,
        "kernel": name,
        "kernel_source_sha16": hashlib.sha256(src).hexdigest()[:16],
        "cv_mlp_mfma_kernel_hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
        "FETCH_SIZE_KB": fetch_kb,
        "WRITE_SIZE_KB": write_kb,
        "source": f"profiles/{tag}_pmc_summary.json (scripts/collect_pmc.sh {tag}: one rocprofv3 --pmc pass per counter over "
                  "bench.py --streams 1, means per dispatch)",
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> doubled "
                      "(MI355X_MICROARCH.md, HBM); Infinity-Cache hits are included in the counter; WRITE_SIZE uncorrected",
        "algorithmic_bytes_per_launch": 15670000.0,
        # matrix-pipe occupancy from the same pass set: every v_mfma_f32_32x32x2_f32 holds the pipe of
        # its SIMD for 64 cycles and executes 4096 flop, so BUSY_CYCLES / 64 = MFMAs executed (the kernel folds the
        # plane-invariant input columns: fewer than the algorithmic count)
        "SQ_VALU_MFMA_BUSY_CYCLES": rec.get("SQ_VALU_MFMA_BUSY_CYCLES"),
        "executed_mfma_flops_per_launch": (rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 64.0 * 4096.0) if rec.get("SQ_VALU_MFMA_BUSY_CYCLES") else None,
        "GRBM_GUI_ACTIVE_per_xcd": (rec["GRBM_GUI_ACTIVE"] / 8.0) if rec.get("GRBM_GUI_ACTIVE") else None,
        "pmc_pass_duration_ns": rec.get("_duration_ns"),
        # the kernel's average under rocprofv3 --kernel-trace --stats (the figure profiles/ reproduces; the
        # HIP-event time bench.py measures live is 1-2.5 % shorter) travels with the record: bench.py prints frac_rocprof from it
        "rocprof_avg_launch_ns": rocprof_ns, "rocprof_calls": rocprof_calls, "rocprof_stats_file": stats_file,
        "note": "bench.py reports this figure only while kernel_source_sha16 equals the hash of "
                "doubletake_amd/csrc/cv_mlp_mfma.hip (otherwise traffic = null: the kernel changed since the counters were collected)",
    }
    json.dump(out, open(os.path.join(REPO, "profiles", "roofline_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
