#!/bin/bash
# clean per-kernel trace of the matching encoder at 8 images and at 1 image (antialiased stem)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r5o}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 8 1; do
  DT_ENC_ONLY=1,$n timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$n -o enc -- python $R/scripts/time_matching_encoder.py > $O/enc$n.json 2>/dev/null
  f=$(find $O/tr$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/encoder_kernel_stats_n$n.csv; rm -rf $O/tr$n
  echo "== n=$n"; cat $O/enc$n.json; cut -d, -f1-4 $O/encoder_kernel_stats_n$n.csv | head -14 | cut -c1-130
done
