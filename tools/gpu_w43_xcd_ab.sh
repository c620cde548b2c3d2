#!/bin/bash
# A/B of the XCD blocking of the F(4x4,3x3) launches' tile order (GANCE_TUNE_W43_XCD = 1: 4 pixel tiles x 8 channel tiles per XCD
# round where the layer has 16 channel tiles; 0: 2 x 16, round 3's): parity subset, per-launch times, FETCH_SIZE of both.
out=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -k "winograd43 or bench_configuration" 2>&1 | tail -2
for m in 1 0 1 0; do
  GANCE_TUNE_W43_XCD=$m timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --print-steps --steps 10 > $out/w43xcd_$m.json 2> $out/w43xcd_$m.steps
  echo "xcd $m: $(python -c "import json; print(json.loads(open('$out/w43xcd_$m.json').read())['value'])") $(grep convV $out/w43xcd_$m.steps | awk '{printf "%s ", $2}')"
done
cd /tmp && export TMPDIR=/tmp
cd "$OLDPWD"
for m in 1 0; do
  GANCE_TUNE_W43_XCD=$m rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/w43xcd_pmc_$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/w43xcd_pmc_$m.err
  python3 tools/pmc_summary.py $(find $out/w43xcd_pmc_$m -name "*counter_collection.csv") > $out/w43xcd_fetch_$m.csv
  rm -rf $out/w43xcd_pmc_$m
  echo "xcd $m FETCH_SIZE (KB):"; grep -E "winograd43" $out/w43xcd_fetch_$m.csv | cut -d, -f1,4,5
done
