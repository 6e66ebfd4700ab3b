#!/bin/bash
# Round 7 probe: the volume kernel with ONE wave per SIMD (DT_MLP_WAVES=4: 256 VGPRs of every SIMD stay free for the conv waves of
# the other keyframes in flight) against the shipped two waves per SIMD, at 4 / 5 / 6 lanes.  Headline bench, same box, interleaved.
out=gpurun_out/${1:-r7d}
mkdir -p $out
for rep in 1 2; do
  for w in 8 4; do
    for st in 4 6; do
      DT_MLP_WAVES=$w timeout 300 python bench.py --streams $st --steps 300 --warmup 30 --no-side-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('waves',$w,'streams',$st,'value %.1f'%d['value'],'ms %.4f'%d['ms_per_step'])" >> $out/waves_probe.txt
    done
  done
done
cat $out/waves_probe.txt
