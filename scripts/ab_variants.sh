#!/bin/bash
# A/B of library variants built by scripts/build_variant.py on the GPU box:
#   bash scripts/ab_variants.sh TAG default pf1 pf2_wpe2 ...      ("default" = the shipped library)
# prints frames/s (two streams), single-stream ms/step, conv stack ms (single stream), dominant kernel ms per variant
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p "$R/gpurun_out/$TAG"
for v in "$@"; do
  if [ "$v" = default ]; then unset DOUBLETAKE_HIP_LIB; else export DOUBLETAKE_HIP_LIB="$R/doubletake_amd/_lib/variants/$v.so"; fi
  python "$R/bench.py" --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline > "$R/gpurun_out/$TAG/bench_$v.json" 2> "$R/gpurun_out/$TAG/bench_$v.err"
  python - "$R/gpurun_out/$TAG/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["single_stream"]
    print(f"{sys.argv[2]:>16s}: {d['value']:7.1f} f/s (2 streams) | single {s['ms_per_step']:.4f} ms  conv {s.get('conv_stack_avg_ms', float('nan')):.4f} ms  volume {s['dominant_kernel_avg_launch_ms']:.4f} ms | launches {d['roofline_conv']['launches']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
