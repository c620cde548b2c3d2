#!/bin/bash
# Timing ablations of the fused up kernel (GANCE_DEBUG_UPFIR / stagger knobs): per-layer times from bench.py's step table.
for cfg in "0 -1" "0 1" "0 8" "1 -1" "2 -1" "4 -1" "6 -1" "12 -1"; do
  set -- $cfg
  GANCE_DEBUG_UPFIR=$1 GANCE_TUNE_UPFIR_PHASES=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --print-steps > /dev/null 2> gpurun_out/ab_$1_$2.steps
  echo "debug=$1 phases=$2: $(grep convTF gpurun_out/ab_$1_$2.steps | awk '{printf "%s ", $2}')"
done
