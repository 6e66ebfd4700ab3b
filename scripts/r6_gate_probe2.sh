#!/bin/bash
# round 6: why is the one-call launch program slower than eager launches with 4 keyframes in flight?  host pacing / lanes / gate
out=gpurun_out/r6c_gate_probe2.txt
: > $out
run() {  # label, env..., -- args
  label=$1; shift
  env "$@" python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-side-legs $ARGS > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - "$label" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:44s}: {d['value']:.1f} f/s  {d['ms_per_step']:.4f} ms/step  host {d['host_issue_ms_per_step']:.3f}  in-region vol {d['roofline']['in_region_avg_launch_ms']:.3f} conv {d['roofline_conv']['in_region_latency_ms']:.3f}")
PY
}
export GPU_MAX_HW_QUEUES=16
ARGS="--launch program --streams 4"; run "program s4 gate off pace 0.85" DT_PIPE_GATE=off DT_BENCH_PACE_MS=0.85
ARGS="--launch program --streams 4"; run "program s4 gate off pace 0.6" DT_PIPE_GATE=off DT_BENCH_PACE_MS=0.6
ARGS="--launch program --streams 4"; run "program s4 gate off pace 1.1" DT_PIPE_GATE=off DT_BENCH_PACE_MS=1.1
ARGS="--launch program --streams 4"; run "program s4 gate off" DT_PIPE_GATE=off
ARGS="--launch program --streams 6"; run "program s6 gate off" DT_PIPE_GATE=off
ARGS="--launch program --streams 8"; run "program s8 gate off" DT_PIPE_GATE=off
ARGS="--launch program --streams 6"; run "program s6 gate volume" DT_PIPE_GATE=volume
ARGS="--launch program --streams 8"; run "program s8 gate volume" DT_PIPE_GATE=volume
ARGS="--launch program --streams 3"; run "program s3 gate off" DT_PIPE_GATE=off
ARGS="--launch program --streams 2"; run "program s2 gate off" DT_PIPE_GATE=off
ARGS="--launch eager --streams 4"; run "eager s4 gate off (hwq 16)" DT_PIPE_GATE=off
ARGS="--launch eager --streams 6"; run "eager s6 gate off (hwq 16)" DT_PIPE_GATE=off
cat $out
