R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4prof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for s in 2 3 4; do
GPU_MAX_HW_QUEUES=8 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$s -o b -- python $R/bench.py --streams $s --steps 50 --warmup 10 --no-cpu-baseline --no-side-legs > $O/b$s.json 2> $O/b$s.err
f=$(find $O/tr$s -name "*kernel_stats.csv" | head -1)
python - "$f" "$O/b$s.json" $s <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
for r in csv.DictReader(open(sys.argv[1])):
    if "cv_mlp_mfma" in r["Name"]:
        print("streams",sys.argv[3],"rocprof avg us %.1f calls %s | line: value %.1f ms/step %.4f in-region %.4f frac %.3f"%(float(r["AverageNs"])/1e3, r["Calls"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
rm -rf $O/tr$s
done
