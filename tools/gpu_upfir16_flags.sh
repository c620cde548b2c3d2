#!/bin/bash
# Per-launch times of the fused up layers (16-channel geometry) under sets of timing-ablation flags (debug build):
#   gpurun -- 'bash tools/gpu_upfir16_flags.sh tag "2 10 66 74 130 202"'
tag=$1
for f in $2; do
  GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_upfir16dbg.so GANCE_DEBUG_UPFIR=$f timeout -k 10 200 python bench.py --steps 5 --warmup 2 \
    --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfir16_${tag}_ab_$f.steps > gpurun_out/upfir16_${tag}_ab_$f.json || exit 1
  echo "flags=$f: $(grep convTF gpurun_out/upfir16_${tag}_ab_$f.steps | awk '{printf "%s ", $2}')"
done
