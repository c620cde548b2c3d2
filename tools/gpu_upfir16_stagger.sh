#!/bin/bash
# Whole-step frames/s and the fused up layers' times with the second resident block of every CU started late (GANCE_TUNE_UPFIR16_STAGGER_US)
for us in 0 10 25 60 0 25; do
  GANCE_TUNE_UPFIR16_STAGGER_US=$us timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --print-steps --steps 10 > gpurun_out/stagger_$us.json 2> gpurun_out/stagger_$us.steps || exit 1
  echo "stagger $us us: $(python -c "import json; print(json.loads(open('gpurun_out/stagger_$us.json').read())['value'])") $(grep convTF gpurun_out/stagger_$us.steps | awk '{printf "%s ", $2}')"
done
