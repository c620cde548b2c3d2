#!/bin/bash
# Parity and timing of the pair form (F(2,2) along x) of the 16-channel fused up kernel against its direct form:
#   gpurun --timeout 900 -- 'bash tools/gpu_upfir16x_check.sh tag'
# writes gpurun_out/upfir16x_<tag>_*.{log,json,steps}
tag=${1:-a}
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -s \
  -k "fused_upsampling or bench_configuration or 1024_frame_matches or 256-3-False" > gpurun_out/upfir16x_${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/upfir16x_${tag}_tests.log
tail -6 gpurun_out/upfir16x_${tag}_tests.log
[ $rc -eq 0 ] || exit 1
for mode in 1 0 1 0; do
  GANCE_TUNE_UPFIR16X=$mode timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --print-steps \
    > gpurun_out/upfir16x_${tag}_mode${mode}.json 2> gpurun_out/upfir16x_${tag}_mode${mode}.steps || exit 1
  echo "pair form $mode: $(python -c "import json,sys; r=json.loads(open('gpurun_out/upfir16x_${tag}_mode${mode}.json').read()); print(r['value'], 'frames/s', r['ms_per_step'], 'ms')") $(grep convTF gpurun_out/upfir16x_${tag}_mode${mode}.steps | awk '{printf "%s ", $2}')"
done
