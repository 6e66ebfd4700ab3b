#!/bin/bash
# heads inside the conv grid: parity, then A/B through the switch DT_HEADS_IN_CONV on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5l}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
for v in 1 0 1 0; do
  DT_HEADS_IN_CONV=$v python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs > $O/bench_h$v.json 2>/dev/null
  python - "$O/bench_h$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["single_stream"]
print("heads_in_conv=%s %.1f f/s | single %.4f ms conv %.4f volume iso %.4f ms | launches %s" % (sys.argv[2], d["value"], s["ms_per_step"], s["conv_stack_avg_ms"], s["dominant_kernel_avg_launch_ms"], d["roofline_conv"]["launches"]))
PY
done 2>&1 | tee $O/heads_ab.txt
for v in 1 0; do
DT_HEADS_IN_CONV=$v DT_CONFIG=cfg4_small DT_MODES=serial,graphs timeout 300 python scripts/time_incremental.py > $O/inc_h$v.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/inc_h$v.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'wall_ms_per_frame' in v: print('heads_in_conv=$v cfg4', k, round(v['wall_ms_per_frame'],3), 'ms/frame')"
done 2>&1 | tee -a $O/heads_ab.txt
