#!/bin/bash
# round 6: lanes per BASELINE shape under launch programs + back-pressure (which default in_flight per config?)
out=gpurun_out/r6n_lanes_all_configs.txt
: > $out
for cfg in cfg2_full cfg3_full_b8 cfg3_small_b8 cfg4_small cfg5_full_d96 cfg5_small_d96; do
for S in 1 2 3 4; do
  python bench.py --config $cfg --streams $S --steps 30 --warmup 6 --no-cpu-baseline --no-side-legs > /tmp/b.json 2> /tmp/b.err || { tail -3 /tmp/b.err; continue; }
  python - $cfg $S >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:>15s} lanes {sys.argv[2]}: {d['value']:7.1f} f/s  {d['ms_per_step']:8.4f} ms/step  host {d['host_issue_ms_per_step']:.3f}")
PY
done; done
cat $out
