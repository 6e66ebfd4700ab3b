#!/bin/bash
# bench.py launch modes side by side on the GPU box:  bash scripts/ab_bench_modes.sh TAG "args1" "args2" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p "$R/gpurun_out/$TAG"
i=0
for a in "$@"; do
  i=$((i+1))
  python "$R/bench.py" --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline $a > "$R/gpurun_out/$TAG/bench_$i.json" 2> "$R/gpurun_out/$TAG/bench_$i.err" || tail -5 "$R/gpurun_out/$TAG/bench_$i.err"
  python - "$R/gpurun_out/$TAG/bench_$i.json" "$a" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d.get("single_stream") or {}
    print(f"{sys.argv[2]:>28s}: {d['value']:7.1f} f/s  {d['ms_per_step']:.4f} ms/step | frac {d['roofline']['frac']:.3f} | single {s.get('ms_per_step', float('nan')):.4f} ms conv {s.get('conv_stack_avg_ms', float('nan')):.4f} | conv(2-stream) {d['roofline_conv']['avg_ms']:.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
