#!/bin/bash
# eager vs --graph (model-level hipGraph replay), 1 and 2 streams:  bash scripts/r4_graph_probe.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/$1; mkdir -p $O
for mode in "--streams 1" "--streams 1 --graph" "--streams 2" "--streams 2 --graph"; do
  timeout 200 python bench.py --steps 60 --warmup 12 $mode --no-cpu-baseline --no-side-legs > $O/b.json 2>$O/b.err || tail -3 $O/b.err
  python - "$O/b.json" "$mode" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-22s %7.1f f/s  %.4f ms/step  launch=%s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["launch"][:40]))
PY
done
