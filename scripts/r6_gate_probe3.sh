#!/bin/bash
# round 6: lanes x gate x host pacing with the one-call launch program (find the default for KeyframePipeline)
out=gpurun_out/r6d_gate_probe3.txt
: > $out
run() {
  label=$1; shift
  env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-legs $ARGS > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - "$label" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:44s}: {d['value']:.1f} f/s  {d['ms_per_step']:.4f} ms/step  host {d['host_issue_ms_per_step']:.3f}  in-region vol {d['roofline']['in_region_avg_launch_ms']:.3f} conv {d['roofline_conv']['in_region_latency_ms']:.3f}")
PY
}
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
ARGS="--launch program --streams 3"; run "program s3 gate off" DT_PIPE_GATE=off
ARGS="--launch program --streams 3"; run "program s3 gate volume" DT_PIPE_GATE=volume
ARGS="--launch program --streams 3"; run "program s3 gate off pace 0.7" DT_PIPE_GATE=off DT_BENCH_PACE_MS=0.7
ARGS="--launch program --streams 3"; run "program s3 gate off pace 1.0" DT_PIPE_GATE=off DT_BENCH_PACE_MS=1.0
ARGS="--launch program --streams 3"; run "program s3 gate off pace 1.2" DT_PIPE_GATE=off DT_BENCH_PACE_MS=1.2
ARGS="--launch program --streams 4"; run "program s4 gate off pace 1.2" DT_PIPE_GATE=off DT_BENCH_PACE_MS=1.2
ARGS="--launch program --streams 4"; run "program s4 gate off pace 1.3" DT_PIPE_GATE=off DT_BENCH_PACE_MS=1.3
ARGS="--launch program --streams 5"; run "program s5 gate off" DT_PIPE_GATE=off
ARGS="--launch eager --streams 3"; run "eager s3 gate off" DT_PIPE_GATE=off
ARGS="--launch eager --streams 4"; run "eager s4 gate off" DT_PIPE_GATE=off
done
cat $out
