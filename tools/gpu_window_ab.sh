#!/bin/bash
# configs[4] on one GPU (three networks switched by the RMS index) with windows of 4 / 8 / 16 pieces per network
for w in 4 8 16 4 8; do
  GANCE_STREAM_WINDOW_PIECES=$w timeout -k 10 200 python bench.py --workload blend --networks 3 --no-cpu-baseline > gpurun_out/window_$w.json 2> gpurun_out/window_$w.err || exit 1
  echo "window $w: $(python -c "import json; r=json.loads(open('gpurun_out/window_$w.json').read().strip().splitlines()[-1]); print(r['value'], r['seconds'])")"
done
