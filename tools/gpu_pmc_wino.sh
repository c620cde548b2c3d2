cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GANCE_TUNE_WINOGRAD=1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_wino -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_wino.err
python3 tools/pmc_summary.py $(find gpurun_out/pmc_wino -name "*counter_collection.csv") > gpurun_out/pmc_wino.csv
rm -rf gpurun_out/pmc_wino
head -12 gpurun_out/pmc_wino.csv
