#!/bin/bash
# paired metadata steps (81 instead of 84 layer-1 steps at K = 7) A/B against -DDT_MLP_PAIR_META=0 on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5j}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
DOUBLETAKE_HIP_LIB=$R/doubletake_amd/_lib/variants/pair0.so timeout 600 python -m pytest tests/test_volume_gpu.py -x -q > $O/pytest_pair0.log 2>&1; echo "pair0 volume tests rc=$?"; tail -2 $O/pytest_pair0.log
for v in default pair0 default pair0; do
  if [ "$v" = default ]; then unset DOUBLETAKE_HIP_LIB; else export DOUBLETAKE_HIP_LIB="$R/doubletake_amd/_lib/variants/$v.so"; fi
  python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/bench_$v.json 2>/dev/null
  python - "$O/bench_$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["single_stream"]
print("%-8s %.1f f/s | single %.4f ms conv %.4f volume iso %.4f ms (frac %.3f)" % (sys.argv[2], d["value"], s["ms_per_step"], s["conv_stack_avg_ms"], s["dominant_kernel_avg_launch_ms"], d["roofline"]["frac"]))
PY
done 2>&1 | tee $O/pair_ab.txt
unset DOUBLETAKE_HIP_LIB
