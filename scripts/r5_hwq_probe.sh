#!/bin/bash
# GPU_MAX_HW_QUEUES with four keyframes in flight (bench.py sets 8 unless the environment says otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${1:-r5hwq}; mkdir -p $O
for q in 4 6 8 12 16 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hw queues $q: %.1f f/s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
done 2>&1 | tee $O/hwq_probe.txt
for m in 48 64 32; do
  DT_CONV_WINO_MIN_BLOCKS=$m python bench.py --steps 100 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wino min blocks $m (pinned for both plans): %.1f f/s  %.4f ms/step single %.4f' % (d['value'], d['ms_per_step'], d['single_stream']['ms_per_step']))"
done 2>&1 | tee -a $O/hwq_probe.txt
