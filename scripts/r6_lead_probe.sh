#!/bin/bash
# round 6: back-pressure (KeyframePipeline max_lead) x lanes with the one-call launch program
out=gpurun_out/r6f_lead_probe.txt
: > $out
run() {
  label=$1; shift
  env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs $ARGS > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; grep -i fault /tmp/b.err >> $out; }
  python - "$label" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:40s}: {d['value']:.1f} f/s  {d['ms_per_step']:.4f} ms/step  host {d['host_issue_ms_per_step']:.3f}  in-region vol {d['roofline']['in_region_avg_launch_ms']:.3f} conv {d['roofline_conv']['in_region_latency_ms']:.3f}")
PY
}
export GPU_MAX_HW_QUEUES=8 DT_PIPE_GATE=off
for rep in 1 2; do
for S in 3 4 5; do for L in $S $((S+1)) $((S+2)) $((S+4)) 0; do
ARGS="--launch program --streams $S"; run "program s$S lead $L" DT_PIPE_LEAD=$L
done; done
ARGS="--launch eager --streams 4"; run "eager s4 lead 0" DT_PIPE_LEAD=0
ARGS="--launch eager --streams 4"; run "eager s4 lead 5" DT_PIPE_LEAD=5
done
cat $out
