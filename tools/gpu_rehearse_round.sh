#!/bin/bash
# Multi-rank rehearsals on a ONE-GPU box (ranks share cuda:0 over gloo: correctness only, the numbers mean nothing):
# the product stream with both drains against the single-rank run, and bench.py's N-rank line with its N-rank extras.
#   gpurun --timeout 1100 -- 'bash tools/gpu_rehearse_round.sh r04'      -> gpurun_out/<tag>_rehearsal_*.log
tag=${1:-rXX}
out=$PWD/gpurun_out
run2() { timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 "${@:2}"; }
export GANCE_REHEARSAL="1024,2160,8,0"
rm -rf /tmp/rehearsal_a && timeout -k 10 300 python tools/rehearse_stream_ranks.py --prepare /tmp/rehearsal_a > $out/${tag}_rehearsal_2ranks_1024_2160.log 2>&1 || exit 1
run2 29517 tools/rehearse_stream_ranks.py --check /tmp/rehearsal_a >> $out/${tag}_rehearsal_2ranks_1024_2160.log 2>&1 || echo "FAILED rank0 drain" >> $out/${tag}_rehearsal_2ranks_1024_2160.log
run2 29518 tools/rehearse_stream_ranks.py --check /tmp/rehearsal_a --drain per-rank >> $out/${tag}_rehearsal_2ranks_1024_2160.log 2>&1 || echo "FAILED per-rank drain" >> $out/${tag}_rehearsal_2ranks_1024_2160.log
grep -E "ranks on one GPU|single rank|FAILED" $out/${tag}_rehearsal_2ranks_1024_2160.log
export GANCE_REHEARSAL="128,200,8,1"
rm -rf /tmp/rehearsal_b && timeout -k 10 300 python tools/rehearse_stream_ranks.py --prepare /tmp/rehearsal_b > $out/${tag}_rehearsal_2ranks_overlay.log 2>&1 || exit 1
run2 29519 tools/rehearse_stream_ranks.py --check /tmp/rehearsal_b >> $out/${tag}_rehearsal_2ranks_overlay.log 2>&1 || echo "FAILED overlay" >> $out/${tag}_rehearsal_2ranks_overlay.log
grep -E "ranks on one GPU|single rank|FAILED" $out/${tag}_rehearsal_2ranks_overlay.log
unset GANCE_REHEARSAL
# bench.py as the driver launches it on a node, two ranks on this one GPU: the weak-scaling line + the N-rank extras (configs[3] both drains, configs[4] + overlay)
GANCE_BENCH_REHEARSAL=1 run2 29520 bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 > $out/${tag}_rehearsal_bench_2ranks.json 2> $out/${tag}_rehearsal_bench_2ranks.err || echo "FAILED bench --gpus 2"
python - <<PY
import json
line = open("$out/${tag}_rehearsal_bench_2ranks.json").read().strip().splitlines()[-1]
r = json.loads(line)
print("bench --gpus 2 (rehearsal):", r["n_gpus"], "ranks,", r["value"], r["unit"], "| extras:", {k: (v.get("value"), v.get("n_gpus"), v.get("error")) for k, v in r.get("extras", {}).items()} if isinstance(r.get("extras"), dict) else r.get("extras"))
PY
GANCE_BENCH_REHEARSAL=1 run2 29521 bench.py --gpus 2 --workload blend --output-side 2160 --drain per-rank --batch 32 > $out/${tag}_rehearsal_blend_2160_per_rank_2ranks.json 2> $out/${tag}_rehearsal_blend_2160_per_rank_2ranks.err || echo "FAILED blend --gpus 2"
tail -c 400 $out/${tag}_rehearsal_blend_2160_per_rank_2ranks.json
