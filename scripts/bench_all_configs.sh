#!/bin/bash
# bench.py at every BASELINE.json shape (bench.py --config ...), one JSON line each -> gpurun_out/TAG/bench_<config>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
mkdir -p "$R/gpurun_out/$TAG"
for c in cfg2_small cfg2_full cfg3_full_b8 cfg3_small_b8 cfg4_small cfg5_full_d96 cfg5_small_d96; do
  python "$R/bench.py" --config $c --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline > "$R/gpurun_out/$TAG/bench_$c.json" 2> "$R/gpurun_out/$TAG/bench_$c.err" || tail -3 "$R/gpurun_out/$TAG/bench_$c.err"
  python - "$R/gpurun_out/$TAG/bench_$c.json" $c <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("single_stream") or {"ms_per_step": d["ms_per_step"], "value": d["value"]}
rc = d["roofline_conv"]
print(f"{sys.argv[2]:>15s}: {d['value']:7.1f} f/s  {d['ms_per_step']:8.4f} ms/step ({d['config']['streams']} streams) | single stream {s.get('ms_per_step', float('nan')):8.4f} ms = {s.get('value', float('nan')):7.1f} f/s | volume frac {d['roofline']['frac']:.3f} | conv {rc['avg_ms_single_stream']:.3f} ms, {rc['launches']} launches, {rc['direct_equivalent_flops_per_step'] / 1e9:.1f} GF, frac {rc['frac_single_stream']:.3f}")
PY
done
