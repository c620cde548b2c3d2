#!/bin/bash
# Per-launch times of the fused up layers under the timing ablations of the debug build (make -C gance_amd/csrc upfirdbg):
# tools/gpu_upfir_ablate.sh 0 1 2 16 32 48 ...   (GANCE_DEBUG_UPFIR flag sets; wrong results by design)
for f in "$@"; do
  GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_upfirdbg.so GANCE_DEBUG_UPFIR=$f timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfir_ab_$f.steps > gpurun_out/upfir_ab_$f.json
  echo "flags=$f: $(grep convTF gpurun_out/upfir_ab_$f.steps | awk '{printf "%s ", $2}')"
done
