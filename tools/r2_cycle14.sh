#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== N=1 default"; timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench_n1.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2_bench_n1.log | cut -c1-600
echo "== N=1 fusion small"; timeout -k 10 200 python bench.py --steps 2 --warmup 1 --height 270 --width 480 --views-per-gpu 8 --fusion --no-cpu-baseline --no-planesweep > gpurun_out/r2_bench_n1f.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2_bench_n1f.log | cut -c1-400; tail -1 gpurun_out/r2_bench_n1f.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['dense_points'])"
echo "== 2 ranks gloo on one device (rehearsal of the N>1 path)"
AMVS_BENCH_BACKEND=gloo AMVS_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --height 270 --width 480 --scene-views 8 > gpurun_out/r2_bench_2rank.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2_bench_2rank.log | cut -c1-500
AMVS_BENCH_BACKEND=gloo AMVS_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --height 270 --width 480 --scene-views 8 --fusion --batches 2 > gpurun_out/r2_bench_2rankf.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2_bench_2rankf.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['scaling'], d['config']['workload'], d.get('dense_points'))"
echo cycle-done
