#!/bin/bash
# Dev experiment: non-persistent vs persistent conv blocks (GANCE_TUNE_PERSIST bit mask over tile ids)
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_synthesis_gpu.py -x -q 2>&1 | tail -3 || exit 1
GANCE_TUNE_PERSIST=0x3fcf timeout -k 10 300 python -m pytest tests/test_synthesis_gpu.py -x -q 2>&1 | tail -3 || exit 1
for mask in 0 0x3fcf 0x0c0f 0x33c0; do
  echo "=== GANCE_TUNE_PERSIST=$mask"
  GANCE_TUNE_PERSIST=$mask timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --print-steps 2>&1 >/dev/null | grep -E "conv(T)?1?[0-9]_|sum of"
done
