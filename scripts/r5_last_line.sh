#!/bin/bash
# the default line and the two shapes whose r5z sweep entries were outliers, on the final library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${1:-r5zz}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - "$O/bench.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; e=d["roofline_encoder"]; ee=d["end_to_end"]
print("bench: %.1f f/s %.4f ms/step host %.3f | volume %.4f ms frac %.3f busy %.3f traffic %s | conv %.4f ms %s launches | single %.4f | enc %.4f ms frac %.3f / one image %.4f | e2e %.1f/%.1f  4 streams %.1f/%.1f | parity %s | cpu %.4f" % (
 d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], r["avg_launch_ms"], r["frac"], r["mfma_busy_frac"], r["traffic"], d["roofline_conv"]["avg_ms"], d["roofline_conv"]["launches"], d["single_stream"]["ms_per_step"],
 e["batched_1_plus_K"]["avg_ms"], e["frac"], e["single_image"]["avg_ms"], ee["cache_off"]["frames_per_s"], ee["cache_on"]["frames_per_s"], ee["streams_4"]["cache_off"]["frames_per_s"], ee["streams_4"]["cache_on"]["frames_per_s"], d["parity"]["ok"], d["cpu_baseline"]["value"]))
PY
STEPS=50 bash scripts/bench_all_configs.sh ${1:-r5zz} 2>&1 | tee $O/bench_all_configs.txt
