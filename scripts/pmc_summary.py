#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into a per-kernel JSON (means per dispatch).

    python scripts/pmc_summary.py OUT.json gpurun_out/pmc_A gpurun_out/pmc_B ...
"""
import collections
import csv
import json
import os
import re
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        f = [os.path.join(d, x) for x in os.listdir(d) if x.endswith("counter_collection.csv")]
        for path in f:
            for r in csv.DictReader(open(path)):
                n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[n]["_duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {}
    for n, c in agg.items():
        if not n.startswith("dt::"):
            continue
        res[n] = {k: sum(v) / len(v) for k, v in c.items()}
        res[n]["dispatches_seen"] = max(len(v) for v in c.values())
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
