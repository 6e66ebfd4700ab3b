#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r5h}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.log
python scripts/profile_step_host.py 300 > $O/host_profile.txt 2>&1; head -22 $O/host_profile.txt | cut -c1-150
for S in 1 4 4; do
  python bench.py --streams $S --steps 120 --warmup 12 --no-cpu-baseline --no-side-legs > $O/host_$S.json 2>/dev/null
  python - "$O/host_$S.json" "$S" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("streams %s: %.1f f/s  %.4f ms/step  host issue %.4f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"], d["host_issue_ms_per_step"]))
PY
done 2>&1 | tee $O/host_issue.txt
DT_CONFIG=cfg4_small DT_MODES=serial,graphs timeout 300 python scripts/time_incremental.py > $O/time_incremental_cfg4.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/time_incremental_cfg4.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'wall_ms_per_frame' in v: print('cfg4', k, round(v['wall_ms_per_frame'],3), 'ms/frame; host', round(v.get('host_issue_ms_per_frame',0),3))"
