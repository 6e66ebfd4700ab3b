#!/bin/bash
out=gpurun_out/r6zz_full_model_knobs.txt
: > $out
run() { label=$1; shift; env "$@" python bench.py --config $CFG --steps 30 --warmup 6 --no-cpu-baseline --no-side-legs $ARGS > /tmp/b.json 2>/dev/null; python - "$label" >> $out <<'PY'
import json,sys
d=json.load(open("/tmp/b.json")); print(f"{sys.argv[1]:44s}: {d['value']:7.1f} f/s  {d['ms_per_step']:8.4f} ms/step")
PY
}
for CFG in cfg2_full cfg3_full_b8; do
for M in 0 3 11 27; do ARGS="--conv-plan $M"; run "$CFG plan mask $M" X=1; done
for WB in 32 48 64 96 160; do ARGS=""; run "$CFG wino min blocks $WB" DT_CONV_WINO_MIN_BLOCKS=$WB; done
done
cat $out
