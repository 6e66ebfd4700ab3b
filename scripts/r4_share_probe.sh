# DT_MLP_OLD_SHARE sweep with the cost-aware plan on (bench.py, volume kernel isolated + two-stream rate)
cd $GRAFT_REPO_ROOT
for sh in ${SHARES:-0.55 0.565 0.58 0.595 0.61}; do
  DT_MLP_OLD_SHARE=$sh timeout 200 python bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['single_stream']
print('share $sh: %.1f f/s | single %.4f ms volume %.4f ms | in-region %.4f' % (d['value'], s['ms_per_step'], s['dominant_kernel_avg_launch_ms'], d['roofline']['avg_launch_ms']))"
done
