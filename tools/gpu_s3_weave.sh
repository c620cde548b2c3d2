#!/bin/bash
# Per-launch times of the split-operand up layers and frames/s under build variants of upfir_split.hip (Makefile: ../libgance_hip_upfirsweave<N>.so,
# ../libgance_hip_upfirslastrow.so, ../libgance_hip_upfirsdepth3.so):  VARIANTS="default upfirslastrow upfirsdepth3 default" tools/gpu_s3_weave.sh
for v in ${VARIANTS:-default upfirsweave1 upfirsweave3 default}; do
  lib=""; [ "$v" != default ] && lib=$PWD/gance_amd/libgance_hip_$v.so
  GANCE_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/weave_$v.steps > gpurun_out/weave_$v.json || exit 1
  echo "variant=$v: $(grep '/s3' gpurun_out/weave_$v.steps | awk '{printf "%s ", $2}') fps $(python3 -c "import json;print(json.load(open('gpurun_out/weave_$v.json'))['value'])")"
done
