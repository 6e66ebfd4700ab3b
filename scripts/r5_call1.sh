#!/bin/bash
# round 5, first GPU call: parity tests, the new default line (4 keyframes in flight), the stream-count probe on THIS box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5a}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - "$O/bench.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("bench: %.1f f/s %.4f ms/step (streams %s, hwq %s) | volume iso %.4f ms frac %.3f in-region %.4f | single %.4f ms conv %.4f | others %s" % (
  d["value"], d["ms_per_step"], d["config"]["streams"], d["config"]["hw_queues"], r["avg_launch_ms"], r["frac"], r["in_region_avg_launch_ms"],
  d["single_stream"]["ms_per_step"], d["single_stream"]["conv_stack_avg_ms"], [(l["streams"], round(l["value"],1)) for l in d["other_stream_counts"]]))
print("e2e:", json.dumps(d["end_to_end"])); print("enc:", json.dumps(d["roofline_encoder"]))
print("dot:", json.dumps(d["roofline_warp_match_dot"]["bound_statement"])[:600])
print("parity:", d["parity"]["ok"], "cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
for cfg in "2 -" "3 -" "4 -" "4 8" "5 8" "6 8"; do
  set -- $cfg
  if [ "$2" = "-" ]; then export GPU_MAX_HW_QUEUES=4; else export GPU_MAX_HW_QUEUES=$2; fi
  python bench.py --streams $1 --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/streams_$1_q$2.json 2>/dev/null
  python - "$O/streams_$1_q$2.json" "$1" "$GPU_MAX_HW_QUEUES" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("streams %s hwq %s: %.1f f/s  %.4f ms/step  in-region volume %.4f ms  iso %.4f ms" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], r["in_region_avg_launch_ms"], r["avg_launch_ms"]))
except Exception as e: print("streams", sys.argv[2], "FAILED", e)
PY
done 2>&1 | tee $O/streams_probe.txt
unset GPU_MAX_HW_QUEUES
timeout 300 python scripts/time_incremental.py > $O/time_incremental.json 2>$O/time_incremental.err; python -c "
import json; d=json.load(open('$O/time_incremental.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'wall_ms_per_frame' in v: print(k, round(v['wall_ms_per_frame'],3), 'ms/frame; host', round(v.get('host_issue_ms_per_frame',0),3))"
