#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5n}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
DOUBLETAKE_HIP_LIB=$R/doubletake_amd/_lib/variants/headw12.so timeout 300 python -m pytest tests/test_networks_gpu.py -x -q -k "small_model or heads" > $O/pytest_headw12.log 2>&1; echo "headw12 tests rc=$?"; tail -2 $O/pytest_headw12.log
for v in default headw12 default headw12; do
  if [ "$v" = default ]; then unset DOUBLETAKE_HIP_LIB; else export DOUBLETAKE_HIP_LIB="$R/doubletake_amd/_lib/variants/$v.so"; fi
  python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/bench_$v.json 2>/dev/null
  python - "$O/bench_$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["single_stream"]
print("%-10s %.1f f/s | single %.4f ms conv %.4f | launches %s" % (sys.argv[2], d["value"], s["ms_per_step"], s["conv_stack_avg_ms"], d["roofline_conv"]["launches"]))
PY
done 2>&1 | tee $O/head_waves_ab.txt
unset DOUBLETAKE_HIP_LIB
python scripts/time_heads.py 2>/dev/null | tail -12 | tee $O/time_heads_default.txt
DOUBLETAKE_HIP_LIB=$R/doubletake_amd/_lib/variants/headw12.so python scripts/time_heads.py 2>/dev/null | tail -12 | tee $O/time_heads_w12.txt
