#!/bin/bash
# Per-launch times of the role-split up layers under the timing ablations of upfir_split_roles.hip:
#   make -C gance_amd/csrc ../libgance_hip_upfirrab<flags>.so ; tools/gpu_upfirr_ablate.sh "" 2 8 23 ...
for v in "$@"; do
  lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_upfirrab$v.so
  GANCE_TUNE_UPFIR_SPLIT_MAXRES=1024 GANCE_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfirr_ab_$v.steps > gpurun_out/upfirr_ab_$v.json || exit 1
  echo "ablate=${v:-none}: $(grep 's3r' gpurun_out/upfirr_ab_$v.steps | awk '{printf "%s ", $2}')"
done
