#!/bin/bash
# round 6: keyframes per launch (batch) x lanes, under launch programs + back-pressure (frames/s of the same model at 640x480)
out=gpurun_out/r6l_batch_probe.txt
: > $out
run() {
  label=$1; shift
  python bench.py --no-cpu-baseline --no-side-legs --steps 40 --warmup 8 "$@" > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; }
  python - "$label" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:30s}: {d['value']:.1f} f/s  {d['ms_per_step']:.4f} ms/step  host {d['host_issue_ms_per_step']:.3f}  conv(single) {d['roofline_conv']['avg_ms']:.3f} ms vol {d['roofline']['avg_launch_ms']:.3f} ms")
PY
}
for S in 1 2 3 4; do run "b2 streams $S" --config cfg2_small_b2 --streams $S; done
for S in 1 2 3; do run "b4 streams $S" --config cfg2_small_b4 --streams $S; done
run "b1 streams 4" --config cfg2_small --streams 4
cat $out
