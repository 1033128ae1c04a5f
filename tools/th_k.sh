#!/bin/bash
# strip height sweep for another patch size:  th_k.sh PATCH "TH list"
for th in $2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --patch $1 --tile-rows $th > gpurun_out/thk.log 2>&1
  tail -1 gpurun_out/thk.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k', $1, 'TH', $th, round(d['value']), d['roofline']['avg_launch_ms'])"
done
