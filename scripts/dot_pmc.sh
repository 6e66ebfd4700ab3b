#!/bin/bash
# Separate PMC summaries of the dot-product kernel at B=1 (cfg2) and B=8 (cfg3): bash scripts/dot_pmc.sh TAG
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
for shape in cfg2 cfg3_b8; do
  bash "$R/scripts/collect_pmc_cmd.sh" ${TAG}_$shape "python scripts/time_dot.py --once --only $shape" "FETCH_SIZE" "WRITE_SIZE" \
     "SQ_INSTS_VALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
  python "$R/scripts/pmc_summary.py" "$R/gpurun_out/${TAG}_${shape}_pmc_summary.json" "$R"/gpurun_out/${TAG}_${shape}_p*/
  find "$R"/gpurun_out/${TAG}_${shape}_p* -name "*.csv" -delete
done
