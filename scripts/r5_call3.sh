#!/bin/bash
# round 5, third GPU call: conv plan objective masks under 4 keyframes in flight
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5c}; O=gpurun_out/$TAG; mkdir -p $O
probe() {  # label, env assignments...
  label=$1; shift
  env "$@" python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/env_$label.json 2>$O/env_$label.err
  python - "$O/env_$label.json" "$label" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); s=d["single_stream"]
    print("%-28s %.1f f/s  %.4f ms/step | single %.4f ms conv %.4f ms volume %.4f | launches %s" % (sys.argv[2], d["value"], d["ms_per_step"], s["ms_per_step"], s["conv_stack_avg_ms"], s["dominant_kernel_avg_launch_ms"], d["roofline_conv"]["launches"]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
{
probe obj0 DT_CONV_OBJ=0
for m in 1 2 3 7 11 15 19 23 31 5 6; do probe obj$m DT_CONV_OBJ=$m; done
probe obj3_wmb48 DT_CONV_OBJ=3 DT_CONV_WINO_MIN_BLOCKS=48
probe obj3_wmb24 DT_CONV_OBJ=3 DT_CONV_WINO_MIN_BLOCKS=24
probe obj7_wmb48 DT_CONV_OBJ=7 DT_CONV_WINO_MIN_BLOCKS=48
probe obj3_nopair DT_CONV_OBJ=3 DT_CONV_PAIR=0
probe obj3_s5 DT_CONV_OBJ=3 X=1
probe obj0_again DT_CONV_OBJ=0
} 2>&1 | tee $O/conv_obj_probe.txt
for m in 3 7; do
  DT_CONV_OBJ=$m python bench.py --streams 5 --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('obj$m streams5', round(d['value'],1))"
  DT_CONV_OBJ=$m python bench.py --streams 3 --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('obj$m streams3', round(d['value'],1))"
done 2>&1 | tee -a $O/conv_obj_probe.txt
timeout 600 python -m pytest tests/test_networks_gpu.py tests/test_model_fullsize_gpu.py -x -q > $O/pytest_obj3.log 2>&1; echo "tests default rc=$?"; tail -2 $O/pytest_obj3.log
DT_CONV_OBJ=31 timeout 600 python -m pytest tests/test_networks_gpu.py tests/test_model_fullsize_gpu.py tests/test_matching_encoder.py -x -q > $O/pytest_obj31.log 2>&1; echo "tests obj31 rc=$?"; tail -2 $O/pytest_obj31.log
