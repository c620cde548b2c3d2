#!/bin/bash
# Timing ablations of the F(4x4,3x3) kernel (results wrong, timing only): per-launch times of the three layers it runs.
for flags in ${W43_FLAGS:-0 4 1 2 3}; do
  GANCE_DEBUG_W43=$flags GANCE_TUNE_WINO43=${W43_MAXRES:-256} timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/w43_ab_$flags.steps >/dev/null
  echo "debug=$flags: $(grep convV gpurun_out/w43_ab_$flags.steps | awk '{printf "%s ", $2}')"
done
