#!/bin/bash
# End-of-round evidence from one gpurun call:  bash scripts/profile_round.sh TAG
#   1. rocprofv3 --kernel-trace --stats of `bench.py --streams 1 --steps 30`      -> gpurun_out/TAG/stats (csv) + rocpd db
#   2. separate --pmc passes (one counter group per pass; never together with --stats-only domains the pool refuses)
#   3. the step timeline of pass 1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -n "$SKIP_TRACE" ] || timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d "$OUT/trace" -o bench -- \
  python "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
# (round 4: instruction-mix and wait counters for the per-kernel account of where SIMD time goes; a counter this gfx950
#  build does not know fails its own pass only)
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM"; do
  n=$(echo $c | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$n" -o pmc -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side-legs --streams 1 > "$OUT/pmc_$n.err" 2>&1
  echo "pmc $c rc=$?"
done
cd "$R"
python scripts/pmc_summary.py "$OUT/pmc_summary.json" "$OUT"/pmc_*/ 2>&1 | tail -1
DB=$(find "$OUT/trace" -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/step_timeline.py "$DB" > "$OUT/step_timeline.txt" 2>&1
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/bench_kernel_stats.csv"
# keep the merge-back small: drop the raw traces
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
ls "$OUT"
