#!/bin/bash
for th in ${TH_LIST:-12 15 16 18 20 22 24 27 30}; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --tile-rows $th > gpurun_out/th_$th.log 2>&1
  tail -1 gpurun_out/th_$th.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($th, round(d['value']), d['roofline']['avg_launch_ms'])"
done
