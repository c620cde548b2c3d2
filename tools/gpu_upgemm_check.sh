#!/bin/bash
# Parity and timing of the scatter form (gemm_forms.hip) of the two smallest up layers against the transposed-conv tiles:
#   gpurun --timeout 900 -- 'bash tools/gpu_upgemm_check.sh tag'
tag=${1:-a}
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -s \
  -k "scatter_form or bench_configuration or layerwise_activations or vector_path or matrix_path_matches" > gpurun_out/upgemm_${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/upgemm_${tag}_tests.log
tail -6 gpurun_out/upgemm_${tag}_tests.log
[ $rc -eq 0 ] || exit 1
for mode in 256 0 256 0; do
  GANCE_TUNE_WINOGEMM=$mode timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --print-steps \
    > gpurun_out/upgemm_${tag}_mode${mode}.json 2> gpurun_out/upgemm_${tag}_mode${mode}.steps || exit 1
  echo "scatter from $mode columns: $(python -c "import json,sys; r=json.loads(open('gpurun_out/upgemm_${tag}_mode${mode}.json').read()); print(r['value'], 'frames/s', r['ms_per_step'], 'ms')") $(grep -E "conv[0-9V]+G?[24]_|finish" gpurun_out/upgemm_${tag}_mode${mode}.steps | awk '{printf "%s ", $2}')"
done
