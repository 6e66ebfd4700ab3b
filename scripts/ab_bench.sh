#!/bin/bash
# A/B of one environment switch on the same GPU box:  bash scripts/ab_bench.sh VAR A B [repeats]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    env "$VAR=$v" timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
