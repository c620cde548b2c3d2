#!/bin/bash
# Batch sweep (frames/s per frames-per-call) under several thresholds of the GEMM forms of the layers at 4^2 ... 16^2
# (GANCE_TUNE_WINOGEMM,GANCE_TUNE_UPGEMM = smallest number of GEMM columns that takes the form; 0 = never):
#   gpurun --timeout 900 -- 'bash tools/gpu_gemm_threshold_sweep.sh 1,2,4,8,16,32,64 256,512 64,128 0,0'
batches=$1
shift
for cfg in "$@"; do
  wino=${cfg%,*}; up=${cfg#*,}
  GANCE_TUNE_WINOGEMM=$wino GANCE_TUNE_UPGEMM=$up python bench.py --no-cpu-baseline --steps 5 --warmup 2 --batch-sweep $batches 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); bs=r['extras']['batch_sweep']['by_batch']
print('wino $wino up $up:', {b: bs[b]['frames_per_s'] for b in bs})"
done
