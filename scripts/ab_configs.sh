#!/bin/bash
# A/B of library variants over several bench configs: bash scripts/ab_configs.sh TAG "cfgA cfgB" default VARIANT ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; CFGS=$2; shift; shift
mkdir -p "$R/gpurun_out/$TAG"
for c in $CFGS; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset DOUBLETAKE_HIP_LIB; else export DOUBLETAKE_HIP_LIB="$R/doubletake_amd/_lib/variants/$v.so"; fi
    python "$R/bench.py" --config $c --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline > "$R/gpurun_out/$TAG/bench_${c}_$v.json" 2> "$R/gpurun_out/$TAG/bench_${c}_$v.err" || tail -3 "$R/gpurun_out/$TAG/bench_${c}_$v.err"
    python - "$R/gpurun_out/$TAG/bench_${c}_$v.json" "$c/$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); s = d.get("single_stream") or {}
    print(f"{sys.argv[2]:>26s}: {d['value']:7.1f} f/s | single {s.get('ms_per_step', float('nan')):8.4f} ms  conv {s.get('conv_stack_avg_ms', float('nan')):8.4f} ms  volume {s.get('dominant_kernel_avg_launch_ms', float('nan')):.4f} ms")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
