#!/bin/bash
# Timing ablations of the 64-channel Winograd kernel (debug build: make -C gance_amd/csrc w64debug).
export GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_w64dbg.so
for dbg in 0 2 32 34 64 66 98 4; do
  GANCE_DEBUG_W64=$dbg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --print-steps > /dev/null 2> gpurun_out/w64ab_$dbg.steps
  echo "debug=$dbg: $(grep convW gpurun_out/w64ab_$dbg.steps | awk '{printf "%s ", $2}')"
done
