for v in "" prev "" prev; do
  lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_$v.so
  GANCE_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/s3ab_$v.steps > gpurun_out/s3ab_$v.json || exit 1
  echo "variant=${v:-new}: $(grep '/s3' gpurun_out/s3ab_$v.steps | awk '{printf "%s ", $2}') fps $(python3 -c "import json;print(json.load(open('gpurun_out/s3ab_$v.json'))['value'])")"
done
