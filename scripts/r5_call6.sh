#!/bin/bash
# host issue time vs GPU time with 2..6 keyframes in flight
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5f}; O=gpurun_out/$TAG; mkdir -p $O
for S in 1 2 3 4 5 6; do
  python bench.py --streams $S --steps 120 --warmup 12 --no-cpu-baseline --no-side-legs > $O/host_$S.json 2>/dev/null
  python - "$O/host_$S.json" "$S" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("streams %s: %.1f f/s  %.4f ms/step  host issue %.4f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"], d["host_issue_ms_per_step"]))
PY
done 2>&1 | tee $O/host_issue.txt
python bench.py --streams 4 --graph --steps 120 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph streams 4: %.1f f/s host %.4f' % (d['value'], d['host_issue_ms_per_step']))" | tee -a $O/host_issue.txt
python bench.py --streams 4 --no-fuse --steps 120 --warmup 12 --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-fuse streams 4: %.1f f/s host %.4f' % (d['value'], d['host_issue_ms_per_step']))" | tee -a $O/host_issue.txt
