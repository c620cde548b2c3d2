#!/bin/bash
# Parity and timing of the role-split form of the split-operand up kernel (upfir_split_roles.hip: matrix waves + vector waves):
#   gpurun --timeout 900 -- 'bash tools/gpu_upfirr_check.sh tag'
# writes gpurun_out/upfirr_<tag>_*.{log,json,steps}
tag=${1:-a}
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -s -k "split_operand" > gpurun_out/upfirr_${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/upfirr_${tag}_tests.log
grep -E "worst layer|rel err|passed|failed|Error|error" gpurun_out/upfirr_${tag}_tests.log | tail -15
[ $rc -eq 0 ] || exit 1
for mode in "1 1024" "0 512" "1 1024"; do
  set -- $mode
  GANCE_TUNE_UPFIR_SPLIT_ROLES=$1 GANCE_TUNE_UPFIR_SPLIT_MAXRES=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --print-steps \
    > gpurun_out/upfirr_${tag}_roles$1.json 2> gpurun_out/upfirr_${tag}_roles$1.steps || exit 1
  echo "roles $1 (split form up to $2): $(python -c "import json,sys; r=json.loads(open('gpurun_out/upfirr_${tag}_roles$1.json').read()); print(r['value'], 'frames/s', r['ms_per_step'], 'ms')") $(grep convTF gpurun_out/upfirr_${tag}_roles$1.steps | awk '{printf "%s ", $2}')"
done
