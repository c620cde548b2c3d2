#!/bin/bash
# PMC passes over the F(4x4,3x3) kernel (and the F(2x2,3x3) one beside it): LDS and issue counters.
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "$OLDPWD"
export GANCE_TUNE_WINO43=${W43_MAXRES:-256}
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/w43_pmc_a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/w43_pmc_a.err &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD --output-format csv -d $out/w43_pmc_b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/w43_pmc_b.err
export PMC_SPLIT="upfir_fused_kernel:4;upfir_fused_pre_kernel:4;winograd64_rgb_kernel:2;winograd43_rgb_kernel:3"
python3 tools/pmc_summary.py $(find $out/w43_pmc_a -name "*counter_collection.csv") > $out/w43_pmc_a.csv
python3 tools/pmc_summary.py $(find $out/w43_pmc_b -name "*counter_collection.csv") > $out/w43_pmc_b.csv
rm -rf $out/w43_pmc_a $out/w43_pmc_b
grep -E "kernel|winograd" $out/w43_pmc_a.csv; grep -E "kernel|winograd" $out/w43_pmc_b.csv
