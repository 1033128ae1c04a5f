#!/bin/bash
# usage: ab.sh variant...   ("base" = in-tree lib)
for v in "$@"; do
  if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.log; continue; }
  tail -1 gpurun_out/ab_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d['roofline']['avg_launch_ms'])"
done
