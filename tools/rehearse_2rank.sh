#!/bin/bash
# Rehearsal of the N = 2 bench path on a one-GPU box: `python bench.py --gpus 2` starts its own two ranks (the
# driver's command shape), both on cuda:0, gloo transport (the real run is one rank per GPU over RCCL); default
# exchange, then --gather-per-iteration; then the same two exchanges on real RCCL with the only topology a one-GPU
# box offers, a ONE-rank nccl group.
set -o pipefail
export AMVS_BENCH_BACKEND=gloo AMVS_BENCH_ONE_DEVICE=1
for extra in "" "--gather-per-iteration"; do
  python bench.py --gpus 2 --steps 2 --warmup 1 --scene-views ${VIEWS:-8} --no-planesweep --no-cpu-baseline $extra 2>&1 | grep -v "amdgpu.ids\|OMP_NUM\|^\*\*\*" || exit 1
done
unset AMVS_BENCH_BACKEND AMVS_BENCH_ONE_DEVICE
for extra in "" "--gather-per-iteration"; do
  AMVS_BENCH_FORCE_EXCHANGE=1 python bench.py --gpus 1 --steps 2 --warmup 1 --scene-views 16 --no-planesweep --no-cpu-baseline $extra 2>&1 | grep -v "amdgpu.ids" || exit 1
done
