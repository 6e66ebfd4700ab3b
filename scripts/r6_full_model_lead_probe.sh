#!/bin/bash
for cfg in cfg2_full cfg3_full_b8; do
for S in 2 3 4; do for L in $S $((S+1)) $((S+3)); do
  DT_PIPE_LEAD=$L python bench.py --config $cfg --streams $S --steps 30 --warmup 6 --no-cpu-baseline --no-side-legs > /tmp/b.json 2>/dev/null
  python - $cfg $S $L <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:>14s} lanes {sys.argv[2]} lead {sys.argv[3]}: {d['value']:7.1f} f/s  {d['ms_per_step']:8.4f} ms/step")
PY
done; done; done
