#!/bin/bash
# PMC counters of the bench step, ONE counter per rocprofv3 pass (FETCH_SIZE + WRITE_SIZE in one pass
# exceeds the hardware's counter capacity and aborts), each pass under a hard timeout.
#   gpurun -- 'bash scripts/collect_pmc.sh TAG FETCH_SIZE WRITE_SIZE ...'
# then:  python scripts/pmc_summary.py profiles/TAG_pmc_summary.json gpurun_out/TAG_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc "$c" --output-format csv -d "$R/gpurun_out/${TAG}_$c" -o pmc -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side-legs --streams 1 > "$R/gpurun_out/${TAG}_$c.err" 2>&1
  echo "$c rc=$?"
done
