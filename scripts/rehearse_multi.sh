#!/bin/bash
# N > 1 rehearsal of bench.py on a ONE-GPU box: N ranks share cuda:0, collectives over gloo (bench.py --rehearse-on-one-gpu).
# Checks that the whole N > 1 path runs and that the assembled frame equals rank 0's own full-frame render.  usage: rehearse_multi.sh [N ...]
set -o pipefail
mkdir -p gpurun_out/rehearse
for n in "${@:-2}"; do
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29700 + n)) \
    bench.py --gpus "$n" --steps 20 --warmup 5 --rehearse-on-one-gpu > "gpurun_out/rehearse/bench_n$n.json" 2> "gpurun_out/rehearse/bench_n$n.err" || { tail -20 "gpurun_out/rehearse/bench_n$n.err"; exit 1; }
  tail -1 "gpurun_out/rehearse/bench_n$n.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('assembled_frame_vs_single_gpu_render'), d['bvh_broadcast_ms'])" || exit 1
done
