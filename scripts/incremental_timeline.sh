#!/bin/bash
# every dispatch of ONE frame of the incremental loop (volume kernel to volume kernel): bash scripts/incremental_timeline.sh TAG
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$R/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
DT_FRAMES=14 DT_MODES=serial timeout -s KILL 200 rocprofv3 --kernel-trace --output-format rocpd -d "$OUT/trace" -o inc -- python "$R/scripts/time_incremental.py" > "$OUT/inc.json" 2> "$OUT/inc.err"
cd "$R"
DB=$(find "$OUT/trace" -name "*.db" | head -1)
python scripts/step_timeline.py "$DB" --step -3 > "$OUT/incremental_frame_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
cat "$OUT/incremental_frame_timeline.txt"
