#!/bin/bash
# PMC counters of an arbitrary command, ONE counter group per rocprofv3 pass, each under a hard timeout.
#   bash scripts/collect_pmc_cmd.sh TAG "python scripts/time_dot.py --once" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_WAVE_CYCLES" ...
# then:  python scripts/pmc_summary.py profiles/TAG_pmc_summary.json gpurun_out/TAG_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; CMD=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  ( cd "$R" && timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/${TAG}_p$i" -o pmc -- $CMD > "$R/gpurun_out/${TAG}_p$i.err" 2>&1 )
  echo "pass $i ($c) rc=$?"
done
