cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f
for cfg in "2:" "3:" "4:" "3:GPU_MAX_HW_QUEUES=8" "4:GPU_MAX_HW_QUEUES=8" "6:GPU_MAX_HW_QUEUES=8"; do
  s=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 200 python bench.py --steps 60 --warmup 12 --streams $s --no-cpu-baseline --no-side-legs > gpurun_out/r4f/b_$s_$e.json 2>/dev/null
  python - "gpurun_out/r4f/b_$s_$e.json" "$s $e" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], "f/s %.1f  ms/step %.4f  volume in-region %.4f ms frac %.3f  conv in-region %.3f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["roofline_conv"]["avg_ms"]))
PY
done
