set -u
mkdir -p gpurun_out/rot
for n in 1 4 8; do
  timeout 240 python bench.py --no-cpu-baseline --no-side-legs --input-sets $n > gpurun_out/rot/sets$n.json 2> gpurun_out/rot/sets$n.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/rot/sets$n.json").read().strip().splitlines()[-1])
print("sets",$n,r["value"],r["ms_per_step"],r["roofline"]["avg_launch_ms"],r["roofline"]["frac"],r.get("single_stream"))
PY
done
timeout 240 python bench.py --no-cpu-baseline --no-side-legs --graph > gpurun_out/rot/graph4.json 2> gpurun_out/rot/graph4.err
tail -c 600 gpurun_out/rot/graph4.json
