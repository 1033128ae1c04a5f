#!/bin/bash
# plane-sweep A/B of libamvs variants ("base" = in-tree)
for v in "$@"; do
  if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --workload planesweep --no-cpu-baseline --steps 10 --warmup 8 > gpurun_out/psab.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/psab.log; continue; }
  tail -1 gpurun_out/psab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d['ms_per_step'])"
done
