#!/bin/bash
# ablation timing of conv_wino4_kernel (variants built by scripts/build_variant.py w4ablN -DDT_W4ABL=N)
for v in ${ABLS:-0 1 2 3 4 8 16 32}; do
  echo "== DT_W4ABL=$v"
  DOUBLETAKE_HIP_LIB=doubletake_amd/_lib/variants/w4abl$v.so DT_W4_SHAPES=short python scripts/wino4_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-100 | tail -3
done
