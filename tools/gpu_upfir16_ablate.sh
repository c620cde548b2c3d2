#!/bin/bash
# The 16-channel fused up kernel (upfir16_fused.hip): parity subset on the product build, per-launch times under the timing
# ablations of the debug build (make -C gance_amd/csrc upfir16dbg; GANCE_DEBUG_UPFIR flag sets, wrong results by design) and
# an occupancy / matrix-pipe counter pass of both geometries.
#   gpurun --timeout 1100 -- 'bash tools/gpu_upfir16_ablate.sh tag "0 2 32 16 1 4 8"'
tag=${1:-a}
flags=${2:-"0 2 32"}
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_synthesis_gpu.py -m gpu -x -q -s \
  -k "fused_upsampling or 512_both or noise_draws" > gpurun_out/upfir16_${tag}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/upfir16_${tag}_tests.log
tail -3 gpurun_out/upfir16_${tag}_tests.log
for f in $flags; do
  GANCE_HIP_LIBRARY=$PWD/gance_amd/libgance_hip_upfir16dbg.so GANCE_DEBUG_UPFIR=$f timeout -k 10 200 python bench.py --steps 5 --warmup 2 \
    --no-cpu-baseline --no-extras --print-steps 2> gpurun_out/upfir16_${tag}_ab_$f.steps > gpurun_out/upfir16_${tag}_ab_$f.json || exit 1
  echo "flags=$f: $(grep convTF gpurun_out/upfir16_${tag}_ab_$f.steps | awk '{printf "%s ", $2}')"
done
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "$OLDPWD"
for mode in 1 0; do
  GANCE_TUNE_UPFIR16=$mode rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $out/upfir16_pmc_$mode -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/upfir16_${tag}_pmc_$mode.err || exit 1
  PMC_SPLIT="upfir_fused_pre_kernel:4;upfir16_fused_pre_kernel:4" python3 tools/pmc_summary.py $(find $out/upfir16_pmc_$mode -name "*counter_collection.csv") > $out/upfir16_${tag}_pmc_$mode.csv
  rm -rf $out/upfir16_pmc_$mode
  grep -E "^kernel|upfir" $out/upfir16_${tag}_pmc_$mode.csv
done
