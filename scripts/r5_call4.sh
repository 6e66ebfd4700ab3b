#!/bin/bash
# round 5, fourth GPU call: validate the throughput plan (full tests + default bench line), encoder kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5d}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - "$O/bench.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("bench: %.1f f/s %.4f ms/step (streams %s, hwq %s, plan %s) | volume iso %.4f ms frac %.3f in-region %.4f | single %.4f ms conv %.4f (plan %s) | others %s" % (
  d["value"], d["ms_per_step"], d["config"]["streams"], d["config"]["hw_queues"], d["config"]["conv_plan_mask"], r["avg_launch_ms"], r["frac"], r["in_region_avg_launch_ms"],
  d["single_stream"]["ms_per_step"], d["single_stream"]["conv_stack_avg_ms"], d["single_stream"]["conv_plan_mask"], [(l["streams"], round(l["value"],1)) for l in d["other_stream_counts"]]))
print("conv:", json.dumps(d["roofline_conv"]))
print("e2e:", json.dumps(d["end_to_end"])); print("enc:", json.dumps(d["roofline_encoder"]))
print("parity:", d["parity"], "cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["whole_frame_s"])
PY
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_enc -o enc -- python $R/scripts/time_matching_encoder.py > $R/$O/time_matching_encoder.json 2>/dev/null )
f=$(find $O/trace_enc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/encoder_kernel_stats.csv; rm -rf $O/trace_enc
cat $O/time_matching_encoder.json; head -30 $O/encoder_kernel_stats.csv | cut -c1-160
