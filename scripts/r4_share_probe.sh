#!/bin/bash
# volume kernel: weighted older/younger span split (DT_MLP_OLD_SHARE) -- parity tests, then isolated + in-bench timings
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
[ -n "$SKIP_TESTS" ] || timeout 300 python -m pytest tests/test_volume_gpu.py tests/test_model_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -2
for sh in ${SHARES:-0.5 0.62 0.66 0.70 0.75}; do
  DT_MLP_OLD_SHARE=$sh timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); s=r['single_stream']; print('share $sh: %.1f f/s  volume in-region %.4f ms  isolated %.4f ms  single %.4f ms' % (r['value'], r['roofline']['avg_launch_ms'], s['dominant_kernel_avg_launch_ms'], s['ms_per_step']))"
done
