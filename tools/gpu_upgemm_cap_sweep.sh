#!/bin/bash
# Batch sweep at small batches under several sizes of the scatter form's product buffer (GANCE_TUNE_UPGEMM_COLUMNS: GEMM columns of a
# 512-channel layer; the up layers of a call that fits take the scatter form instead of the two-pass transposed conv):
#   gpurun --timeout 900 -- 'bash tools/gpu_upgemm_cap_sweep.sh 1,2,3,4,5,6,7 4096 16384'
batches=$1
shift
for cap in "$@"; do
  GANCE_TUNE_UPGEMM_COLUMNS=$cap python bench.py --no-cpu-baseline --steps 5 --warmup 2 --batch-sweep $batches 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); bs=r['extras']['batch_sweep']['by_batch']
print('buffer $cap columns:', {b: bs[b]['frames_per_s'] for b in bs}, {b: (bs[b]['forms'].get('64x64_up'), bs[b]['forms'].get('128x128_up'), bs[b]['forms'].get('256x256_up')) for b in bs})"
done
