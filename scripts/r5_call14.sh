#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${1:-r5y}; mkdir -p $O
timeout 900 python -m pytest tests/test_volume_gpu.py tests/test_model_fullsize_gpu.py -x -q 2>&1 | tail -2
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr -o b -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1 > /dev/null 2>&1 )
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); grep "mlp_plan\|cv_mlp_mfma" $f | cut -d, -f1-4 | cut -c1-120; rm -rf $O/tr
python bench.py --steps 80 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench %.1f f/s single %.4f ms' % (d['value'], d['single_stream']['ms_per_step']))"
