#!/bin/bash
# round 6: launch mode x volume gate of parallel.KeyframePipeline at the headline config (4 keyframes in flight)
out=gpurun_out/r6b_gate_probe.txt
: > $out
for rep in 1 2; do
for m in program eager; do for g in volume off; do
  DT_PIPE_GATE=$g python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-side-legs --launch $m > /tmp/b.json 2> /tmp/b.err || tail -5 /tmp/b.err
  python - "$m" "$g" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"launch {sys.argv[1]:8s} gate {sys.argv[2]:7s}: {d['value']:.1f} f/s  {d['ms_per_step']:.4f} ms/step  host {d['host_issue_ms_per_step']:.3f}  single {d['single_stream']['value']:.1f} (host {d['single_stream']['host_issue_ms_per_step']:.3f})  in-region vol {d['roofline']['in_region_avg_launch_ms']:.3f} conv {d['roofline_conv']['in_region_latency_ms']:.3f}")
PY
done; done; done
cat $out
