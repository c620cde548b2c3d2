#!/bin/bash
# Parity and timing of the 16-channel geometry of the fused up kernel (upfir16_fused.hip) against the 32-channel one, and
# optionally its timing ablations (debug build: make -C gance_amd/csrc upfir16dbg):
#   gpurun --timeout 900 -- 'bash tools/gpu_upfir16_check.sh tag ["0 2 32"]'
# writes gpurun_out/upfir16_<tag>_*.{log,json,steps}
tag=${1:-a}
flags=${2:-}
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -s \
  -k "fused_upsampling or 512_both or every_term_on_default or noise_draws" > gpurun_out/upfir16_${tag}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/upfir16_${tag}_tests.log
tail -4 gpurun_out/upfir16_${tag}_tests.log
for mode in 1 0; do
  GANCE_TUNE_UPFIR16=$mode timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --print-steps \
    > gpurun_out/upfir16_${tag}_mode${mode}.json 2> gpurun_out/upfir16_${tag}_mode${mode}.steps || exit 1
  echo "mode $mode: $(python -c "import json,sys; r=json.loads(open('gpurun_out/upfir16_${tag}_mode${mode}.json').read()); print(r['value'], 'frames/s', r['ms_per_step'], 'ms')") $(grep convTF gpurun_out/upfir16_${tag}_mode${mode}.steps | awk '{printf "%s ", $2}')"
done
for f in $flags; do
  GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_upfir16dbg.so GANCE_DEBUG_UPFIR=$f timeout -k 10 200 python bench.py --steps 5 --warmup 2 \
    --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfir16_${tag}_ab_$f.steps > gpurun_out/upfir16_${tag}_ab_$f.json || exit 1
  echo "flags=$f: $(grep convTF gpurun_out/upfir16_${tag}_ab_$f.steps | awk '{printf "%s ", $2}')"
done
