#!/bin/bash
# LDS counters of the F(4x4,3x3) kernel under a library variant: tools/gpu_pmc_w43b.sh <variant or ""> 
out=$PWD/gpurun_out
v=$1; lib=""; [ -n "$v" ] && lib=$PWD/gance_amd/libgance_hip_$v.so
export GANCE_HIP_LIBRARY=$lib
cd /tmp && export TMPDIR=/tmp
cd "$OLDPWD"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/w43b_pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/w43b_pmc.err
python3 tools/pmc_summary.py $(find $out/w43b_pmc -name "*counter_collection.csv") > $out/w43b_${v:-default}_pmc.csv
rm -rf $out/w43b_pmc
grep -E "^kernel|winograd43_rgb" $out/w43b_${v:-default}_pmc.csv
