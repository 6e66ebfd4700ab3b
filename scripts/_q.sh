R=$PWD
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/gpurun_out/r2s -o b -- python bench.py --steps 30 --warmup 5 --streams 1 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/r2s.err)
cd $R; db=$(find gpurun_out/r2s -name "*.db" | head -1); python scripts/step_timeline.py $db > gpurun_out/r2s_step_timeline.txt; rm -rf gpurun_out/r2s/*/*.db
cat gpurun_out/r2s_step_timeline.txt
