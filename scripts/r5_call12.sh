#!/bin/bash
# the driver's own commands on a fresh box: smoke(), then the exact BENCH command with its wall clock
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5p}; O=gpurun_out/$TAG; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | tail -4
python - "$O/bench_driver_cmd.json" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.strip()]
assert len(lines)==1, len(lines)
d=json.loads(lines[0]); r=d["roofline"]
print("driver cmd: %.1f f/s %.4f ms/step host %.3f | volume %.4f ms frac %.3f traffic %s | conv %.4f ms | parity %s | cpu %.4f f/s (%s cores)" % (d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], r["avg_launch_ms"], r["frac"], r["traffic"], d["roofline_conv"]["avg_ms"], d["parity"]["ok"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
print("keys:", sorted(d.keys()))
PY
