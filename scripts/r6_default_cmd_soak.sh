#!/bin/bash
# round 6: soak of the driver's own command (all side legs, launch programs): every run must exit 0 with a finite line
out=gpurun_out/r6zz_default_cmd_soak.txt
: > $out
fails=0
for i in $(seq ${REPS:-60}); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
  rc=$?
  if [ $rc -ne 0 ] || ! python -c "import json;d=json.load(open('/tmp/b.json'));assert d['value']>0" 2>/dev/null; then
    fails=$((fails+1)); echo "== FAIL run $i rc=$rc" >> $out; grep -v amdgpu.ids /tmp/b.err | tail -6 >> $out
  else
    python -c "import json;d=json.load(open('/tmp/b.json'));print(round(d['value'],1))" >> /tmp/vals.txt
  fi
done
python - >> $out <<'PY'
import numpy as np
v=np.loadtxt('/tmp/vals.txt')
print(f"values: n={v.size} min {v.min():.1f} median {np.median(v):.1f} max {v.max():.1f} frames/s")
PY
echo "$fails / ${REPS:-60} failed" >> $out
cat $out
