#!/bin/bash
# Parity and timing of the experiment GANCE_TUNE_GEMM_BF16X6 (GEMM forms of the layers at 4^2 ... 16^2 on the bf16 matrix cores, split operands):
#   gpurun --timeout 900 -- 'bash tools/gpu_gemm_bf16_check.sh tag'
tag=${1:-a}
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -s -k "scatter_form or stress_network_256" > gpurun_out/gemmbf16_${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/gemmbf16_${tag}_tests.log
tail -6 gpurun_out/gemmbf16_${tag}_tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
for mode in 0 1 2; do
  GANCE_TUNE_GEMM_BF16X6=$mode timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --print-steps \
    > gpurun_out/gemmbf16_${tag}_mode${mode}.json 2> gpurun_out/gemmbf16_${tag}_mode${mode}.steps || exit 1
  echo "bf16x6 $mode: $(python -c "import json,sys; r=json.loads(open('gpurun_out/gemmbf16_${tag}_mode${mode}.json').read()); print(r['value'], 'frames/s', r['ms_per_step'], 'ms')") $(grep -E "conv[TV]G" gpurun_out/gemmbf16_${tag}_mode${mode}.steps | awk '{printf "%s ", $2}')"
done
export GANCE_TUNE_GEMM_BF16X6=2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gemmbf16_${tag}_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/gemmbf16_${tag}_trace.err
cp $(find gpurun_out/gemmbf16_${tag}_trace -name "*kernel_stats.csv" | head -1) gpurun_out/gemmbf16_${tag}_kernel_stats.csv && rm -rf gpurun_out/gemmbf16_${tag}_trace
grep -E "tile_gemm|pack|gather|finish" gpurun_out/gemmbf16_${tag}_kernel_stats.csv | cut -c1-160
