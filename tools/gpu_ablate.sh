#!/bin/bash
# Dev experiment: per-launch table of the conv layers under the debug ablation flags of conv_mfma.hip
# (1 = no stores, 2 = no DMA after the first chunk, 4 = no MFMA, 64 = no XCD remap); needs `make debug`
export GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_dbg.so
for flags in ${@:-0 1 2 3}; do
  echo "=== GANCE_DEBUG_CONV=$flags"
  GANCE_DEBUG_CONV=$flags timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --print-steps 2>&1 >/dev/null | grep -E "conv(T)?1[0-6]_|conv8_|convT[79]_|sum of"
done
