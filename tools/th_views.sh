#!/bin/bash
# strip height x views per GPU: separates tail effects from per-strip effects
for v in 16 32; do for th in 24 40 54; do
  timeout -k 10 250 python bench.py --no-cpu-baseline --tile-rows $th --views-per-gpu $v > gpurun_out/thv.log 2>&1
  tail -1 gpurun_out/thv.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views', $v, 'TH', $th, round(d['value']), d['roofline']['avg_launch_ms'])"
done; done
