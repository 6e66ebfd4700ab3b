#!/bin/bash
# round 6: plan objective masks / back-pressure / lanes under the launch-program + back-pressure regime (driver command and 60 steps)
out=gpurun_out/r6k_knob_probe.txt
: > $out
run() {
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-side-legs $ARGS > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; grep -i fault /tmp/b.err >> $out; }
  python - "$label" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:46s}: {d['value']:.1f} f/s  {d['ms_per_step']:.4f} ms/step  host {d['host_issue_ms_per_step']:.3f} wait {d['host_wait_ms_per_step']:.3f}  single {d['single_stream']['value']:.1f}")
PY
}
for rep in 1 2; do
for L in 4 5 6; do ARGS="--steps 20 --warmup 5"; run "driver cmd lead $L" DT_PIPE_LEAD=$L; done
for M in 0 3 11 15 27; do ARGS="--steps 60 --warmup 10 --conv-plan $M"; run "60 steps plan mask $M" X=1; done
for WB in 32 48 64 96; do ARGS="--steps 60 --warmup 10"; run "60 steps wino min blocks $WB" DT_CONV_WINO_MIN_BLOCKS=$WB; done
ARGS="--steps 60 --warmup 10 --launch eager"; run "60 steps eager" X=1
done
cat $out
