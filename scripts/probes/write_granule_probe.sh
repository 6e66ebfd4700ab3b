#!/bin/bash
# gpurun -- 'bash scripts/probes/write_granule_probe.sh TAG'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r7e}
out=$R/gpurun_out/$TAG
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/wgp $R/scripts/probes/write_granule_probe.hip || exit 1
for c in WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; do
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o pmc -- /tmp/wgp > $out/$c.log 2>&1
  echo "$c rc=$?"
done
python - <<PY
import csv, glob, collections
for c in ("WRITE_SIZE", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    print(c, {k: round(sum(v) / len(v), 1) for k, v in sorted(acc.items())})
PY
