#!/bin/bash
# Per-launch times of the F(4x4,3x3) layers under library variants: tools/gpu_w43_variants.sh "" w43win1 w43win2 ...
for v in "$@"; do
  lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_$v.so
  GANCE_HIP_LIBRARY=$lib GANCE_TUNE_WINO43=${W43_MAXRES:-256} timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/w43_var_$v.steps > gpurun_out/w43_var_$v.json
  echo "variant=${v:-default}: $(grep convV gpurun_out/w43_var_$v.steps | awk '{printf "%s ", $2}') sum $(grep 'sum of' gpurun_out/w43_var_$v.steps | awk '{print $4}') fps $(python3 -c "import json;print(json.load(open('gpurun_out/w43_var_$v.json'))['value'])")"
done
