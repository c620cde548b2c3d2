#!/bin/bash
# PMC passes over the role-split up kernel (upfir_split_roles.hip): wait / issue counters, LDS, L2 hits.  usage: tools/gpu_pmc_upfirr.sh <tag> [ablation flags of a built variant]
tag=${1:-a}
out=$PWD/gpurun_out
lib=""; [ -n "$2" ] && lib=$PWD/gance_amd/libgance_hip_upfirrab$2.so
export GANCE_HIP_LIBRARY=$lib
cd /tmp && export TMPDIR=/tmp
cd "$OLDPWD"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/upfirr_pmc_a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/upfirr_pmc_a.err &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/upfirr_pmc_b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/upfirr_pmc_b.err &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/upfirr_pmc_c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/upfirr_pmc_c.err
export PMC_SPLIT="upfirr_fused_pre_kernel:4;upfirr_fused_pre_noise_kernel:4;upfirs_fused_pre_kernel:4"
for p in a b c; do
  f=$(find $out/upfirr_pmc_$p -name "*counter_collection.csv" 2>/dev/null)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f > $out/upfirr_${tag}_pmc_$p.csv
  rm -rf $out/upfirr_pmc_$p
  grep -E "^kernel|upfir" $out/upfirr_${tag}_pmc_$p.csv
  tail -3 $out/upfirr_pmc_$p.err
done
