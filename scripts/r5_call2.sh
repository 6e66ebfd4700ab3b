#!/bin/bash
# round 5, second GPU call: full parity tests, space-partition probe, conv-plan switches under 4 keyframes in flight, incremental at cfg4
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5b}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python scripts/partition_probe.py 2>&1 | tee $O/partition_probe.txt | tail -60
probe() {  # label, env assignments...
  label=$1; shift
  env "$@" python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/env_$label.json 2>/dev/null
  python - "$O/env_$label.json" "$label" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); s=d["single_stream"]
    print("%-28s %.1f f/s  %.4f ms/step | single %.4f ms conv %.4f ms volume %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], s["ms_per_step"], s["conv_stack_avg_ms"], s["dominant_kernel_avg_launch_ms"]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
probe default X=1
probe conv_split4 DT_CONV_SPLIT=4
probe wino_ksplit1 DT_WINO_KSPLIT=1
probe split4_ksplit1 DT_CONV_SPLIT=4 DT_WINO_KSPLIT=1
probe no_kparts DT_CONV_KPARTS=1 DT_WINO_KPARTS=0
probe no_tail DT_CONV_TAIL_SPLIT=0
probe no_kparts_no_tail DT_CONV_KPARTS=1 DT_WINO_KPARTS=0 DT_CONV_TAIL_SPLIT=0
probe head_split_all DT_HEAD_SPLIT_MAX_TILES=100000
probe default_again X=1
} 2>&1 | tee $O/conv_env_probe.txt
DT_CONFIG=cfg4_small DT_MODES=serial,lookahead,graphs timeout 300 python scripts/time_incremental.py > $O/time_incremental_cfg4.json 2>$O/time_incremental_cfg4.err; python -c "
import json; d=json.load(open('$O/time_incremental_cfg4.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'wall_ms_per_frame' in v: print('cfg4', k, round(v['wall_ms_per_frame'],3), 'ms/frame; host', round(v.get('host_issue_ms_per_frame',0),3), 'hint', round(v['hint_time_ms_median'],3), 'model', round(v['model_time_ms_median'],3))"
