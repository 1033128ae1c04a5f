#!/bin/bash
# plane sweep: strip height x plane chunk
for th in 16 24 32; do for ch in 4 8 16; do
  AMVS_SWEEP_TILE_ROWS=$th AMVS_SWEEP_CHUNK=$ch timeout -k 10 200 python bench.py --workload planesweep --no-cpu-baseline > gpurun_out/ps.log 2>&1
  tail -1 gpurun_out/ps.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TH', $th, 'chunk', $ch, round(d['value']), d['ms_per_step'])"
done; done
timeout -k 10 200 python bench.py --workload planesweep --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['value']), d['ms_per_step'], d['config'])"
