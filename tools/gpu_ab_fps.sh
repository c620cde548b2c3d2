#!/bin/bash
# Dev experiment: whole-step frames/s of library / tuning variants, interleaved, 3 rounds.
# usage: gpu_ab_fps.sh "lib:ENV=val,ENV=val" ...   (lib = prev | new)
variants=("$@")
[ ${#variants[@]} -eq 0 ] && variants=("prev:" "new:")
for round in 1 2 3; do
  for variant in "${variants[@]}"; do
    lib=${variant%%:*}; envs=${variant#*:}
    (
      if [ "$lib" = prev ]; then export GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_prev.so; fi
      IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
      fps=$(timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])")
      echo "round $round $variant $fps"
    )
  done
done
