#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5r}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
bash scripts/r5_enc_trace.sh $TAG 2>&1 | grep -v "^\"void at\|rocclr" | tail -30
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.json 2>/dev/null; python - $O/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); e=d["roofline_encoder"]; ee=d["end_to_end"]
print("bench %.1f f/s | single conv %.4f | encoder batched %.4f ms frac %.3f, single image %.4f ms | e2e 1 stream %.1f / %.1f, 4 streams %.1f / %.1f" % (d["value"], d["single_stream"]["conv_stack_avg_ms"], e["batched_1_plus_K"]["avg_ms"], e["frac"], e["single_image"]["avg_ms"], ee["cache_off"]["frames_per_s"], ee["cache_on"]["frames_per_s"], ee["streams_4"]["cache_off"]["frames_per_s"], ee["streams_4"]["cache_on"]["frames_per_s"]))
PY
