#!/bin/bash
# end-of-round evidence in one gpurun call:  bash scripts/r6_final_evidence.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/pytest_gpu.log
bash scripts/profile_round.sh ${TAG}p > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
cp gpurun_out/${TAG}p/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/${TAG}p/step_timeline.txt $O/step_timeline.txt 2>/dev/null
cp gpurun_out/${TAG}p/pmc_summary.json profiles/${TAG}_pmc_summary.json 2>/dev/null && python scripts/make_roofline_traffic.py $TAG profiles/${TAG}_bench_kernel_stats.csv > /dev/null && cp profiles/roofline_traffic.json $O/roofline_traffic.json && cp profiles/${TAG}_pmc_summary.json profiles/${TAG}_bench_kernel_stats.csv $O/
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_default -o bench -- python $R/bench.py --no-cpu-baseline --no-side-legs > $R/$O/bench_under_default_trace.json 2> /dev/null )
f=$(find $O/trace_default -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats_default_cmd.csv; rm -rf $O/trace_default
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench (driver cmd) rc=$?"
python - "$O/bench.json" "$O/bench_driver_cmd.json" <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); r=d["roofline"]
    print("%s: %.1f f/s %.4f ms/step host %.3f | volume %.4f ms frac %.3f rocprof %s busy %s traffic %s | single %s | parity %s | cpu %.3f f/s" % (f, d["value"], d["ms_per_step"], d["host_issue_ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("frac_rocprof"), r.get("mfma_busy_frac"), r.get("traffic"), d["single_stream"]["ms_per_step"], d["parity"]["ok"], d["cpu_baseline"]["value"]))
PY
STEPS=50 bash scripts/bench_all_configs.sh $TAG 2>&1 | tee $O/bench_all_configs.txt
for cfg in cfg2_small cfg4_small; do
DT_CONFIG=$cfg DT_MODES=serial,lookahead,graphs,programs,programs+lookahead timeout 300 python scripts/time_incremental.py > $O/time_incremental_$cfg.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/time_incremental_$cfg.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'wall_ms_per_frame' in v: print('$cfg', k, round(v['wall_ms_per_frame'],3), 'ms/frame; host', round(v['host_issue_ms_per_frame'],3))"
done
timeout 300 python scripts/time_two_pass.py > $O/time_two_pass.json 2>/dev/null; cat $O/time_two_pass.json; echo
bash scripts/r6_batch_probe.sh > /dev/null 2>&1; cp gpurun_out/r6l_batch_probe.txt $O/batch_probe.txt; cat $O/batch_probe.txt
python bench.py --force-dist --no-cpu-baseline --no-side-legs > $O/bench_forcedist.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_forcedist.json')); print('force-dist', round(d['value'],1), 'f/s ranks', d['config']['ranks_seen'])"
python bench.py --force-dist --tsdf-mode slab --tsdf-res 0.02 --no-cpu-baseline --no-side-legs > $O/bench_forcedist_slab.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_forcedist_slab.json')); print('force-dist slab 0.02', round(d['value'],1), 'f/s')"
python bench.py --launch eager --no-cpu-baseline --no-side-legs > $O/bench_eager.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_eager.json')); print('eager launches', round(d['value'],1), 'f/s host', round(d['host_issue_ms_per_step'],3))"
