#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats, bench.py --streams 1) of library variants:
#   bash scripts/kernel_stats_variants.sh TAG "kernel-name regex" default NAME ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift; PAT=$1; shift
O="$R/gpurun_out/$TAG"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset DOUBLETAKE_HIP_LIB; else export DOUBLETAKE_HIP_LIB="$R/doubletake_amd/_lib/variants/$v.so"; fi
  rm -rf "$O/tr_$v"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tr_$v" -o b -- \
    python "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1 > "$O/tr_$v.json" 2> "$O/tr_$v.err"
  f=$(find "$O/tr_$v" -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  [ -n "$f" ] && python - "$f" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
  [ -n "$f" ] && cp "$f" "$O/kernel_stats_$v.csv"
  rm -rf "$O/tr_$v"
done
