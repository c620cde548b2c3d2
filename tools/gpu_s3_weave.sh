for v in "" upfirsweave1 upfirsweave3 upfirsweave4 ""; do
  lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_$v.so
  GANCE_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/weave_$v.steps > gpurun_out/weave_$v.json || exit 1
  echo "variant=${v:-default}: $(grep '/s3' gpurun_out/weave_$v.steps | awk '{printf "%s ", $2}') fps $(python3 -c "import json;print(json.load(open('gpurun_out/weave_$v.json'))['value'])")"
done
