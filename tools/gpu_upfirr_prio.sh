for v in "" prio0 prio2; do
  lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_upfirr$v.so
  GANCE_TUNE_UPFIR_SPLIT_MAXRES=1024 GANCE_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfirr_p_$v.steps > gpurun_out/upfirr_p_$v.json || exit 1
  echo "variant=${v:-default}: $(grep 's3r' gpurun_out/upfirr_p_$v.steps | awk '{printf "%s ", $2}')"
done
