#!/bin/bash
# usage: th_ab.sh variant "TH list"
v=$1; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
for th in $2; do
  timeout -k 10 250 python bench.py --no-cpu-baseline --tile-rows $th > gpurun_out/thab.log 2>&1
  tail -1 gpurun_out/thab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'TH', $th, round(d['value']), d['roofline']['avg_launch_ms'])"
done
