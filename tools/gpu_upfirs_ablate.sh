#!/bin/bash
# Per-launch times of the split-operand up layers under the timing ablations of upfir_split.hip:
#   make -C gance_amd/csrc ../libgance_hip_upfirsab<flags>.so ; tools/gpu_upfirs_ablate.sh "" 2 3 6 ...
for v in "$@"; do
  lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_upfirsab$v.so
  GANCE_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfirs_ab_$v.steps > gpurun_out/upfirs_ab_$v.json
  echo "ablate=${v:-none}: $(grep 's3' gpurun_out/upfirs_ab_$v.steps | awk '{printf "%s ", $2}')"
done
