#!/bin/bash
# round 5, fifth GPU call: fast geometry (v_rcp + Newton, u - 0.5) A/B against -DDT_FAST_GEOM=0 on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5e}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
for v in default geom0 default geom0; do
  if [ "$v" = default ]; then unset DOUBLETAKE_HIP_LIB; else export DOUBLETAKE_HIP_LIB="$R/doubletake_amd/_lib/variants/$v.so"; fi
  python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/bench_$v.json 2>/dev/null
  python - "$O/bench_$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["single_stream"]; dot=d["roofline_warp_match_dot"]
print("%-8s %.1f f/s | single %.4f ms conv %.4f volume iso %.4f ms (frac %.3f) | dot B1 %.4f ms valu_frac %.3f  B8 %.4f ms" % (sys.argv[2], d["value"], s["ms_per_step"], s["conv_stack_avg_ms"], s["dominant_kernel_avg_launch_ms"], d["roofline"]["frac"], dot["avg_launch_ms"], dot["valu_frac"], dot["batch8_512x384"]["avg_launch_ms"]))
PY
done 2>&1 | tee $O/geom_ab.txt
unset DOUBLETAKE_HIP_LIB
