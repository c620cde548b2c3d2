#!/bin/bash
# Round profile set (run under gpurun from the repo root): rocprofv3 kernel-trace stats, three
# separate PMC passes (never combined with a trace domain other than --kernel-trace), bench lines.
#   usage: tools/gpu_profile_round.sh <tag>      -> files under gpurun_out/<tag>_*
tag=${1:-rXX}
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
repo=$OLDPWD
cd "$repo"
set -x
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_trace.err &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/${tag}_pmc_fetch.err &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/${tag}_pmc_write.err &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/${tag}_pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/${tag}_pmc_sq.err &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $out/${tag}_pmc_valu -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/${tag}_pmc_valu.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_audio_trace -- python3 bench.py --workload blend --no-cpu-baseline > $out/${tag}_blend_under_rocprof.json 2> $out/${tag}_audio_trace.err
python3 bench.py --steps 20 --warmup 3 --all-terms --no-extras --no-cpu-baseline > $out/${tag}_bench_all_terms.json 2> /dev/null
python3 tools/gpu_one_frame_latency.py > $out/${tag}_one_frame_latency.json 2> $out/${tag}_one_frame_steps.txt
set +x
for d in trace pmc_fetch pmc_write pmc_sq; do find $out/${tag}_$d -name "*.csv" | head -5; done
python3 tools/pmc_summary.py $(find $out/${tag}_pmc_fetch -name "*counter_collection.csv") > $out/${tag}_pmc_fetch_size.csv
python3 tools/pmc_summary.py $(find $out/${tag}_pmc_write -name "*counter_collection.csv") > $out/${tag}_pmc_write_size_clock.csv
python3 tools/pmc_summary.py $(find $out/${tag}_pmc_sq -name "*counter_collection.csv") > $out/${tag}_pmc_sq.csv
python3 tools/pmc_summary.py $(find $out/${tag}_pmc_valu -name "*counter_collection.csv") > $out/${tag}_pmc_valu.csv
cp $(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# the product's blend job (files on disk -> frames on the host): every kernel incl. the audio -> latent ones
cp $(find $out/${tag}_audio_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_audio_kernel_stats.csv
# (the record takes the workload and the per-launch table from a bench line: a short one first, the full one after it)
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --print-steps > $out/${tag}_bench.json 2> $out/${tag}_steps.txt
python3 tools/make_traffic_record.py ${tag} $out && cp profiles/traffic_latest.json $out/${tag}_traffic_latest.json
# the un-profiled default line last: its roofline.traffic then comes from THIS set's record
python3 bench.py --steps 20 --warmup 3 --print-steps > $out/${tag}_bench.json 2> $out/${tag}_steps.txt
# the raw traces are large: keep the summaries only
rm -rf $out/${tag}_trace $out/${tag}_audio_trace $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_sq $out/${tag}_pmc_valu
head -5 $out/${tag}_kernel_stats.csv; cat $out/${tag}_bench.json | head -c 600
